#!/bin/bash
# After `gpurun -- 'bash profiles/run_round6_profiles.sh'`: turn the raw gpurun_out/r6p_* files into the committed summaries of profiles/.
set -eu
cd "$(dirname "$0")/.."
for c in headline cfg2 cfg3 cfg4 cfg5; do python3 profiles/summarise.py r6p_$c round6 $c r6p_headline; done
cp gpurun_out/r6p_bench_full.json profiles/round6_bench_line_full.json
cp gpurun_out/r6p_pytest.log profiles/round6_gpu_pytest_durations.log
cp gpurun_out/r6p_filters.txt profiles/round6_filters.txt
cp gpurun_out/r6p_filters_timeline.txt profiles/round6_filters_timeline.txt
cp gpurun_out/r6p_opt_stage_trace.txt profiles/round6_opt_stage_trace.txt
cp gpurun_out/r6p_batch_scaling.txt profiles/round6_batch_scaling.txt
grep -v "Warn\|amdgpu.ids" gpurun_out/r6p_service_latency.txt > profiles/round6_service_latency.txt
grep "^{" gpurun_out/r6p_bench_nccl1_packed.json | tail -1 > profiles/round6_bench_line_nccl_one_rank_packed.json
grep "^{" gpurun_out/r6p_bench_2rank.json | tail -1 > profiles/round6_bench_line_2rank_one_gpu.json
cp gpurun_out/r6p_stage_traces.txt profiles/round6_seq_stage_traces.txt
{ cat profiles/round6_service_latency.txt; echo; echo "# host side of the one-pose service call (profiles/probe_host_timing.py, -DFPE_HOST_TIMING build):"; cat gpurun_out/r6p_host_timing.txt; } > profiles/round6_service_latency.tmp && mv profiles/round6_service_latency.tmp profiles/round6_service_latency.txt
