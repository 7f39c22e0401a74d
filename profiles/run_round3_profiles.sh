#!/bin/bash
# Round-3 evidence at HEAD: per configuration kernel stats + counters (bit-window kernels), the direct kernels' stats for
# comparison, the full bench line of the headline (cpu_baseline, ingest, open loop, D2H-inclusive rate), the open-loop
# kernel's counter bytes, and the -m gpu suite with durations.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r3p_pytest.log 2>&1; echo "pytest rc=$?"
bash profiles/collect.sh r3p_headline headline full > /dev/null 2>&1
for c in cfg2 cfg3 cfg4 cfg5; do bash profiles/collect.sh r3p_$c $c lite > /dev/null 2>&1; done
for c in headline cfg3 cfg4 cfg5; do bash profiles/collect.sh r3p_${c}_direct $c stats --no-bits > /dev/null 2>&1; done
python3 bench.py > gpurun_out/r3p_bench_full.json 2> gpurun_out/r3p_bench_full.err; echo "full bench rc=$?"
FPE_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3p_bench_2rank.json 2> gpurun_out/r3p_bench_2rank.err; echo "2-rank rc=$?"
# open-loop kernel: counter bytes of search_legs_kernel (the full bench runs it when extras are on)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3p_ol_fetch -o ol -- python3 bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3p_ol_write -o ol -- python3 bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3p_ol_stats -o ol -- python3 bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1
# the producer's filter chain (N3): per-kernel split by resolution
rm -rf gpurun_out/prof_filters
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_filters -o f -- python3 profiles/probe_filters.py > gpurun_out/probe_filters.txt 2>&1
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"; do
  d=gpurun_out/prof_filters_pmc_$(echo $grp | tr ' ' '_')
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o f -- python3 profiles/probe_filters.py > /dev/null 2>&1
done
# residency / arbitration study of the one-wavefront-per-pose kernels (profiling build scratch/libfpe_trace.so:
# -DFPE_TRACE -DFPE_TRACE_ALL_BLOCKS, built by `EXTRA_DEFS=-DFPE_TRACE_ALL_BLOCKS bash profiles/build_trace.sh` before the push)
if [ -f scratch/libfpe_trace.so ]; then
  for c in cfg3 cfg5 headline; do FPE_LIB=scratch/libfpe_trace.so python3 profiles/probe_residency.py $c 4096; done > gpurun_out/r3p_residency.txt 2>&1
  for b in 1024 2048 3072 4096; do for c in cfg3 cfg5; do
    python3 bench.py --config $c --batch $b --steps 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c poses $b (= %d wavefronts per SIMD) kernel_ms %.4f' % ($b // 1024, l['roofline']['kernel_ms']))"; done; done >> gpurun_out/r3p_residency.txt 2>&1
fi
# stage traces (s_memtime stamps, profiling build): where a gait cycle / a leg search goes, per kernel family
if [ -f scratch/libfpe_trace.so ]; then
  { echo "== headline (plan_bits_kernel<2,true>)"; FPE_LIB=scratch/libfpe_trace.so python3 profiles/trace_stages.py headline 4096 2>&1 | grep -v "Warn\|amdgpu.ids" | head -24;
    echo "== cfg4 (plan_bits_kernel<3,false>)"; FPE_LIB=scratch/libfpe_trace.so python3 profiles/trace_stages_generic.py cfg4 32768 2>&1 | grep -v "Warn\|amdgpu.ids";
    echo "== cfg3 (plan_bits_seq_kernel<1,2>)"; FPE_LIB=scratch/libfpe_trace.so python3 profiles/trace_stages_seq.py cfg3 4096 2>&1 | grep -v "Warn\|amdgpu.ids\|RuntimeWarning\|ret = \|_methods";
    echo "== cfg5 (plan_bits_seq_kernel<2,3>)"; FPE_LIB=scratch/libfpe_trace.so python3 profiles/trace_stages_seq.py cfg5 4096 2>&1 | grep -v "Warn\|amdgpu.ids\|RuntimeWarning\|ret = \|_methods"; } > gpurun_out/r3p_stage_traces.txt
fi
# service-shaped calls: latency split (plan kernel / opt track)
python3 profiles/probe_service_latency.py > gpurun_out/r3p_service_latency.txt 2>&1
ls gpurun_out | grep r3p_ | wc -l
