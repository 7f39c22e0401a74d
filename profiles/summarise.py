#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of profiles/collect.sh (gpurun_out/<tag>_*) into the committed summaries:
   profiles/<round>_kernel_stats.csv, <round>_pmc_hbm.json, <round>_sq_counters.json, pmc_traffic.json.
   usage: python profiles/summarise.py r1c round1"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PLAN = "plan_chained_kernel"
CANON = "canonicalise_layer_kernel"


def counters(dirname, kernel_substr):
    """mean per-launch value of every counter collected in `dirname` for kernels matching the name"""
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(OUT, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
                n[row["Counter_Name"]] += 1
    return {k: acc[k] / n[k] for k in acc}, dict(n)


# 1. kernel stats (short kernel names only: the torch RNG kernel's name is kilobytes long)
src = glob.glob(os.path.join(OUT, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(src)))
with open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"), "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        name = r["Name"] if len(r["Name"]) < 200 else r["Name"][:120] + "...(truncated)"
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
plan_row = [r for r in rows if PLAN in r["Name"]][0]

# 2. PMC: all passes
allc, alln = {}, {}
for d in sorted(glob.glob(os.path.join(OUT, f"{tag}_pmc*"))):
    if os.path.isdir(d):
        c, n = counters(os.path.basename(d), PLAN)
        allc.update(c)
        alln.update(n)
cf, _ = counters(f"{tag}_calib_fetch", CANON)
cw, _ = counters(f"{tag}_calib_write", CANON)
known = 4000 * 4000 * 4  # bytes read == bytes written per canonicalise launch
fetch_factor = known / (cf["FETCH_SIZE"] * 1024.0)
write_factor = known / (cw["WRITE_SIZE"] * 1024.0)
fetch_b = allc["FETCH_SIZE"] * 1024.0 * fetch_factor
write_b = allc["WRITE_SIZE"] * 1024.0 * write_factor
bench = json.loads(open(os.path.join(OUT, f"{tag}_bench.json")).read().strip().splitlines()[-1])
alg = bench["roofline"]["algorithmic_bytes_per_foothold"] * bench["config"]["footholds_per_step"]
hbm = {
    "command": "bash profiles/collect.sh %s  (rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline; one run per counter group)" % tag,
    "kernel": plan_row["Name"][:80],
    "workload": bench["config"]["workload"],
    "launches_averaged": alln.get("FETCH_SIZE"),
    "raw_per_launch": {k: allc[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum") if k in allc},
    "calibration": {
        "kernel": "fpe::canonicalise_layer_kernel on one 4000x4000 f32 layer (column-major -> row-major): 64,000,000 B read + 64,000,000 B written per launch (profiles/calib.py)",
        "FETCH_SIZE_KB": cf["FETCH_SIZE"], "WRITE_SIZE_KB": cw["WRITE_SIZE"],
        "fetch_factor": fetch_factor, "write_factor": write_factor,
        "note": "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of this known-traffic kernel (MI355X_MICROARCH.md HBM section), WRITE_SIZE is exact",
    },
    "headline": {
        "hbm_bytes_per_launch": fetch_b + write_b,
        "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (fetch_b + write_b) / alg,
        "l2_hit_rate": allc["TCC_HIT_sum"] / (allc["TCC_HIT_sum"] + allc["TCC_MISS_sum"]) if "TCC_HIT_sum" in allc else None,
        "note": "the 8 MB map is cache resident; each XCD L2 pulls its own copy of the touched lines through the fabric (Infinity-Cache hits included), so these are fabric bytes, an upper bound of HBM bytes",
    },
}
json.dump(hbm, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm.json"), "w"), indent=1)
json.dump({"headline": {"hbm_bytes_per_launch": fetch_b + write_b}}, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)

# 3. SQ counters per wavefront
waves = allc.get("SQ_WAVES")
sq = {"unit": "per wavefront, per launch (counter / SQ_WAVES); *_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* are quad-cycles",
      "kernel": plan_row["Name"][:80], "waves_per_launch": waves,
      "per_wave": {k: v / waves for k, v in sorted(allc.items()) if k.startswith("SQ_") and waves}}
pw = sq["per_wave"]
if "SQ_ACTIVE_INST_VALU" in pw and "SQ_WAVE_CYCLES" in pw:
    sq["derived"] = {"valu_active_fraction_of_wave_cycles": pw["SQ_ACTIVE_INST_VALU"] / pw["SQ_WAVE_CYCLES"],
                     "wait_fraction_of_wave_cycles": pw.get("SQ_WAIT_ANY", 0) / pw["SQ_WAVE_CYCLES"]}
json.dump(sq, open(os.path.join(ROOT, "profiles", f"{rnd}_sq_counters.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(ROOT, "profiles", f"{rnd}_bench_line.json"), "w"), indent=1)
print("plan kernel avg ns (rocprof --stats):", plan_row["AverageNs"], "calls", plan_row["Calls"])
print("bench kernel_ms:", bench["roofline"]["kernel_ms"], "value", bench["value"])
print("traffic/launch MB:", (fetch_b + write_b) / 1e6, "alg MB", alg / 1e6, "L2 hit", hbm["headline"]["l2_hit_rate"])
print("per wave:", {k: round(v, 1) for k, v in pw.items()})
