#!/bin/bash
# fabric traffic per kernel of the filter chain on one map: profiles/collect_filters_traffic.sh MAPIDX
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FPE_PROBE_MAP=$1
rm -rf gpurun_out/r5traf
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/r5traf/$grp -o f -- python3 profiles/probe_filters.py > /dev/null 2>&1
done
python3 - "$1" <<'PY'
import collections, csv, glob, json, sys
cal = json.load(open("profiles/round4_headline_counters.json"))["calibration"]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r5traf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"map {sys.argv[1]}: fabric bytes per launch (FETCH_SIZE / WRITE_SIZE in KiB x the calibration of profiles/round4_headline_counters.json: {cal['fetch_factor']:.3f} / {cal['write_factor']:.3f})")
for k, c in cnt.items():
    m = {n: sum(x) / len(x) for n, x in c.items()}
    print(f"{k}: read {m['FETCH_SIZE']*1024*cal['fetch_factor']/1e6:.1f} MB written {m['WRITE_SIZE']*1024*cal['write_factor']/1e6:.1f} MB  (launches {len(c['FETCH_SIZE'])})")
PY
