#!/bin/bash
# After `gpurun -- 'bash profiles/run_round3_profiles.sh; python3 profiles/probe_batch_scaling.py > gpurun_out/r3p_batch_scaling.txt'`:
# turn the raw gpurun_out/r3p_* files into the committed summaries of profiles/.
set -eu
cd "$(dirname "$0")/.."
for c in headline cfg2 cfg3 cfg4 cfg5; do python3 profiles/summarise.py r3p_$c round3 $c r3p_headline; done
cp gpurun_out/r3p_bench_full.json profiles/round3_bench_line_full.json
cp gpurun_out/r3p_pytest.log profiles/round3_gpu_pytest_durations.log
[ -f gpurun_out/r3p_batch_scaling.txt ] && cp gpurun_out/r3p_batch_scaling.txt profiles/round3_batch_scaling.txt
[ -f gpurun_out/r3p_residency.txt ] && grep -v "Warn\|amdgpu.ids" gpurun_out/r3p_residency.txt > profiles/round3_residency.txt
[ -f gpurun_out/r3p_stage_traces.txt ] && cp gpurun_out/r3p_stage_traces.txt profiles/round3_stage_traces.txt
[ -f gpurun_out/r3p_service_latency.txt ] && grep -v "Warn\|amdgpu.ids" gpurun_out/r3p_service_latency.txt > profiles/round3_service_latency.txt
[ -f gpurun_out/r3p_bench_2rank.json ] && cp gpurun_out/r3p_bench_2rank.json profiles/round3_bench_line_2rank_one_gpu.json
[ -d gpurun_out/prof_filters ] && python3 profiles/summarise_filters.py > profiles/round3_filters.txt
python3 - <<'PY'
import csv, glob, json
def mean_counter(d, name, kern):
    acc = []
    for f in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name'] and r['Counter_Name'] == name: acc.append(float(r['Counter_Value']))
    return sum(acc) / len(acc), len(acc)
fs, n = mean_counter('r3p_ol_fetch', 'FETCH_SIZE', 'search_legs_kernel')
ws, _ = mean_counter('r3p_ol_write', 'WRITE_SIZE', 'search_legs_kernel')
st = glob.glob('gpurun_out/r3p_ol_stats/**/*kernel_stats.csv', recursive=True)[0]
avg = float([r for r in csv.DictReader(open(st)) if 'search_legs_kernel' in r['Name']][0]['AverageNs'])
cal = json.load(open('profiles/round3_headline_counters.json'))['calibration']
ff, wf = cal['fetch_factor'], cal['write_factor']
b = fs * 1024 * ff + ws * 1024 * wf
import os
o = json.load(open('profiles/round3_open_loop_counters.json')) if os.path.exists('profiles/round3_open_loop_counters.json') else json.load(open('profiles/round2_open_loop_counters.json'))
o.update({"launches_averaged": n, "kernel_avg_ns_rocprof_stats": avg, "FETCH_SIZE_KiB": fs, "WRITE_SIZE_KiB": ws, "fetch_factor": ff, "write_factor": wf,
          "fabric_bytes_per_launch": b, "achieved_GBps_by_counter_bytes": b / avg, "frac_of_8TBps_by_counter_bytes": b / avg / 8000.0})
o["by_convention_508B_per_query"] = {"bytes": 66584576, "frac": 66584576 / avg / 8000.0}
json.dump(o, open('profiles/round3_open_loop_counters.json', 'w'), indent=1)
print("open loop", avg, b / avg / 8000.0)
for c in ('headline', 'cfg3', 'cfg4', 'cfg5'):
    f = glob.glob(f'gpurun_out/r3p_{c}_direct_stats/*kernel_stats.csv')
    if f:
        for r in csv.DictReader(open(f[0])):
            if 'plan_' in r['Name']: print(c, 'direct', r['Name'][:50], r['AverageNs'])
PY
