#!/bin/bash
# After `gpurun -- 'bash profiles/run_round5_profiles.sh'`: turn the raw gpurun_out/r5p_* files into the committed summaries of profiles/.
set -eu
cd "$(dirname "$0")/.."
for c in headline cfg2 cfg3 cfg4 cfg5; do python3 profiles/summarise.py r5p_$c round5 $c r5p_headline; done
cp gpurun_out/r5p_bench_full.json profiles/round5_bench_line_full.json
cp gpurun_out/r5p_pytest.log profiles/round5_gpu_pytest_durations.log
cp gpurun_out/r5p_filters.txt profiles/round5_filters.txt
cp gpurun_out/r5p_filters_timeline.txt profiles/round5_filters_timeline.txt
cp gpurun_out/r5p_opt_stage_trace.txt profiles/round5_opt_stage_trace.txt
cp gpurun_out/r5p_batch_scaling.txt profiles/round5_batch_scaling.txt
grep -v "Warn\|amdgpu.ids" gpurun_out/r5p_service_latency.txt > profiles/round5_service_latency.txt
grep "^{" gpurun_out/r5p_bench_nccl1_packed.json | tail -1 > profiles/round5_bench_line_nccl_one_rank_packed.json
grep "^{" gpurun_out/r5p_bench_2rank.json | tail -1 > profiles/round5_bench_line_2rank_one_gpu.json
