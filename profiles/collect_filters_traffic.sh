#!/bin/bash
# fabric traffic per kernel of the filter chain on one map, PER MODE: profiles/collect_filters_traffic.sh MAPIDX
# (one rocprofv3 --pmc pass per counter and mode; FPE_PROBE_MODE makes every launch of a pass belong to one chain)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FPE_PROBE_MAP=$1
rm -rf gpurun_out/r6traf
for mode in trav layers; do
  export FPE_PROBE_MODE=$mode
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/r6traf/$mode/$grp -o f -- python3 profiles/probe_filters.py > /dev/null 2>&1
  done
done
unset FPE_PROBE_MODE
python3 - "$1" <<'PY'
import collections, csv, glob, json, sys
cal = json.load(open("profiles/round4_headline_counters.json"))["calibration"]
maps = ((1000, 0.02), (2000, 0.01), (2000, 0.005))
rows, res = maps[int(sys.argv[1])]
cells = rows * rows
print(f"map {sys.argv[1]} ({rows} x {rows} @ {res} m): fabric bytes per launch, per mode (FETCH_SIZE / WRITE_SIZE in KiB x the calibration of profiles/round4_headline_counters.json: {cal['fetch_factor']:.3f} / {cal['write_factor']:.3f})")
for mode, moved in (("trav", 20), ("layers", 48)):
    # bytes the chain has to MOVE per cell: first launch reads elevation, writes step_height (8 B); second reads elevation and step_height
    # and writes traversability (12 B) — or all seven remaining layers (4 + 4 + 7 * 4 = 36 B... plus the first launch's 8 = 44; step_height
    # is written by the first launch only) -> 20 B traversability-only, 8 + 8 + 28 = 44 B with every layer
    moved = 20 if mode == "trav" else 44
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r6traf/{mode}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "filter_" in r["Kernel_Name"]:
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
                cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tot_r = tot_w = 0.0
    for k, c in cnt.items():
        m = {n: sum(x) / len(x) for n, x in c.items()}
        rd, wr = m["FETCH_SIZE"] * 1024 * cal["fetch_factor"], m["WRITE_SIZE"] * 1024 * cal["write_factor"]
        tot_r += rd
        tot_w += wr
        print(f"  {mode}: {k}: read {rd/1e6:.1f} MB written {wr/1e6:.1f} MB  (launches {len(c['FETCH_SIZE'])})")
    print(f"  {mode}: chain total {(tot_r+tot_w)/1e6:.1f} MB = {(tot_r+tot_w)/(moved*cells):.2f} x the {moved} B/cell the chain moves ({moved*cells/1e6:.1f} MB)")
PY
