#!/bin/bash
# Round-6 evidence at HEAD (one gpurun call): the -m gpu suite with durations, per configuration kernel stats + counters (headline
# full with the FETCH/WRITE calibration, cfg-2..5 lite), the full bench line of the headline, the one-rank RCCL lines, the two-rank
# line, and the producer's filter chain: per-kernel times, instruction counters and fabric traffic per map, both chains.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r6p_pytest.log 2>&1; echo "pytest rc=$?"
bash profiles/collect.sh r6p_headline headline full > /dev/null 2>&1
for c in cfg2 cfg3 cfg4 cfg5; do bash profiles/collect.sh r6p_$c $c lite > /dev/null 2>&1; done
python3 bench.py > gpurun_out/r6p_bench_full.json 2> gpurun_out/r6p_bench_full.err; echo "full bench rc=$?"
FPE_BENCH_FORCE_NCCL=1 python3 bench.py --gpus 1 --steps 20 --warmup 4 --blocks 5 --no-cpu-baseline --no-extras --gather-every 8 > gpurun_out/r6p_bench_nccl1_packed.json 2> gpurun_out/r6p_bench_nccl1_packed.err; echo "nccl world-1 packed rc=$?"
FPE_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 10 --warmup 2 --blocks 5 --no-cpu-baseline > gpurun_out/r6p_bench_2rank.json 2> gpurun_out/r6p_bench_2rank.err; echo "2-rank rc=$?"
# the filter chain, map by map (0: 1000^2 @ 2 cm, 1: 2000^2 @ 1 cm, 2: 2000^2 @ 0.5 cm)
{
  echo "# Round 6: the producer's filter chain (fpe_traversability_device), default parameters, synthetic rough terrain (seed 5); MI355X."
  echo "# per map: chain times by HIP events (both chains), per-kernel times (rocprofv3 --kernel-trace), instruction counters per wavefront and"
  echo "# fabric traffic per launch (rocprofv3 --pmc, one pass per group; FETCH_SIZE x the calibration of the headline profile).  Commands:"
  echo "# profiles/probe_filters.py, profiles/collect_filters_counters.sh <map> tree, profiles/collect_filters_traffic.sh <map>"
  for m in 0 1 2; do
    export FPE_PROBE_MAP=$m
    python3 profiles/probe_filters.py 2>/dev/null | grep " m:"
    rm -rf gpurun_out/r6p_filt_stats
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6p_filt_stats -o f -- python3 profiles/probe_filters.py > /dev/null 2>&1
    python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r6p_filt_stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("  kernels: " + ", ".join(f"{k} {sum(v)/len(v):.1f} us x{len(v)}" for k, v in acc.items()))
PY
    unset FPE_PROBE_MAP
    bash profiles/collect_filters_counters.sh $m tree 2>/dev/null | grep -v "^==" | sed 's/^/  counters: /'
    bash profiles/collect_filters_traffic.sh $m 2>/dev/null | grep -v "^map" | sed 's/^/  fabric (per mode):/'
  done
  echo "# flat ground (every disc of equal elevations: the z axis from the step height, no walk) and a tilted plane with noise:"
  python3 profiles/probe_filters_terrain.py 2>/dev/null | grep "ms per chain"
} > gpurun_out/r6p_filters.txt 2>&1
bash profiles/collect_filters_timeline.sh > /dev/null 2>&1
bash profiles/collect_opt_trace.sh > /dev/null 2>&1
python3 profiles/probe_batch_scaling.py 2>&1 | grep "n_cycles" > gpurun_out/r6p_batch_scaling.txt
python3 profiles/probe_service_latency.py > gpurun_out/r6p_service_latency.txt 2>&1
# stage traces of the one-wavefront-per-pose kernels at HEAD (profiling build scratch/libfpe_trace.so: bash profiles/build_trace.sh before the push)
if [ -f scratch/libfpe_trace.so ]; then
  { echo "== cfg3 (plan_bits_seq_kernel<1,2>)"; FPE_LIB=scratch/libfpe_trace.so python3 profiles/trace_stages_seq.py cfg3 4096 2>&1 | grep -v "Warn\|amdgpu.ids\|RuntimeWarning\|ret = \|_methods";
    echo "== cfg5 (plan_bits_seq_kernel<2,3>)"; FPE_LIB=scratch/libfpe_trace.so python3 profiles/trace_stages_seq.py cfg5 4096 2>&1 | grep -v "Warn\|amdgpu.ids\|RuntimeWarning\|ret = \|_methods"; } > gpurun_out/r6p_stage_traces.txt
fi
# where a one-pose service call spends its time on the host (measurement build scratch/libfpe_ht.so: -DFPE_HOST_TIMING), polled and stream-wait forms
if [ -f scratch/libfpe_ht.so ]; then
  for poll in 1 0; do echo "== service_poll $poll"; FPE_PROBE_POLL=$poll FPE_LIB=scratch/libfpe_ht.so python3 profiles/probe_host_timing.py 2>&1 | grep -v "amdgpu.ids"; done > gpurun_out/r6p_host_timing.txt
fi
ls gpurun_out | grep r6p_ | wc -l
