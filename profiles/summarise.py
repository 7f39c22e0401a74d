#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of profiles/collect.sh (gpurun_out/<tag>_*) into the committed summaries of one
configuration:
   profiles/<round>_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats (every kernel of the bench command)
   profiles/<round>_<cfg>_counters.json      PMC: fabric traffic (FETCH/WRITE, calibrated), L2 hit rate, SQ counters per wavefront
   profiles/<round>_<cfg>_bench_line.json    the bench.py JSON line of the same command
   profiles/pmc_traffic.json                 {cfg: {"hbm_bytes_per_launch": ...}} read back by bench.py (roofline.traffic)
usage: python profiles/summarise.py <tag> <round> <cfg> [calib-tag]
The FETCH_SIZE / WRITE_SIZE correction factors come from the calibration runs of <calib-tag> (default: <tag>): a kernel
with known traffic (canonicalise_layer_kernel on a 4000x4000 layer), as /opt/skills/guides/MI355X_MICROARCH.md prescribes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, rnd, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
calib_tag = sys.argv[4] if len(sys.argv) > 4 else tag
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PLAN_NAMES = ("plan_bits_kernel", "plan_bits_seq_kernel", "plan_chained_kernel", "plan_sequential_kernel")
CANON = "canonicalise_layer_kernel"


def counters(dirname, names):
    """mean per-launch value of every counter collected in `dirname` for kernels matching one of the names"""
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(OUT, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row["Kernel_Name"] for k in names):
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
                n[row["Counter_Name"]] += 1
    return {k: acc[k] / n[k] for k in acc}, dict(n)


# 1. kernel stats (short kernel names only: the torch RNG kernel's name is kilobytes long)
src = glob.glob(os.path.join(OUT, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(src)))
with open(os.path.join(ROOT, "profiles", f"{rnd}_{cfg}_kernel_stats.csv"), "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        name = r["Name"] if len(r["Name"]) < 200 else r["Name"][:120] + "...(truncated)"
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
plan_rows = [r for r in rows if any(k in r["Name"] for k in PLAN_NAMES)]
plan_row = max(plan_rows, key=lambda r: float(r["TotalDurationNs"]))

# 2. PMC: all passes
allc, alln = {}, {}
for d in sorted(glob.glob(os.path.join(OUT, f"{tag}_pmc*"))):
    if os.path.isdir(d):
        c, n = counters(os.path.basename(d), PLAN_NAMES)
        allc.update(c)
        alln.update(n)
cf, _ = counters(f"{calib_tag}_calib_fetch", (CANON,))
cw, _ = counters(f"{calib_tag}_calib_write", (CANON,))
known = 4000 * 4000 * 4  # bytes read == bytes written per canonicalise launch
fetch_factor = known / (cf["FETCH_SIZE"] * 1024.0) if "FETCH_SIZE" in cf else 2.0
write_factor = known / (cw["WRITE_SIZE"] * 1024.0) if "WRITE_SIZE" in cw else 1.0
bench = json.loads(open(os.path.join(OUT, f"{tag}_bench.json")).read().strip().splitlines()[-1])
alg = bench["roofline"]["algorithmic_bytes_per_foothold"] * bench["config"]["footholds_per_step"]
out = {
    "command": f"bash profiles/collect.sh {tag} {cfg} <mode>  (rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python3 bench.py "
               f"--config {cfg} --no-cpu-baseline --no-extras; one run per counter group)",
    "kernel": plan_row["Name"][:90],
    "workload": bench["config"]["workload"],
    "kernel_avg_ns_rocprof_stats": float(plan_row["AverageNs"]),
    "kernel_calls": int(plan_row["Calls"]),
    "bench_kernel_ms_hip_events": bench["roofline"]["kernel_ms"],
    "roofline_frac_bench": bench["roofline"]["frac"],
    "launches_averaged": alln.get("FETCH_SIZE"),
    "raw_per_launch": {k: allc[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum") if k in allc},
    "calibration": {
        "kernel": "fpe::canonicalise_layer_kernel on one 4000x4000 f32 layer (column-major -> row-major): 64,000,000 B read + 64,000,000 B written per launch (profiles/calib.py)",
        "tag": calib_tag, "FETCH_SIZE_KB": cf.get("FETCH_SIZE"), "WRITE_SIZE_KB": cw.get("WRITE_SIZE"),
        "fetch_factor": fetch_factor, "write_factor": write_factor,
        "note": "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of this known-traffic kernel (MI355X_MICROARCH.md HBM section), WRITE_SIZE is exact",
    },
}
if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
    fetch_b = allc["FETCH_SIZE"] * 1024.0 * fetch_factor
    write_b = allc["WRITE_SIZE"] * 1024.0 * write_factor
    out["traffic"] = {
        "hbm_bytes_per_launch": fetch_b + write_b,
        "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (fetch_b + write_b) / alg,
        "l2_hit_rate": allc["TCC_HIT_sum"] / (allc["TCC_HIT_sum"] + allc["TCC_MISS_sum"]) if "TCC_HIT_sum" in allc else None,
        "note": "fabric bytes of the 8 XCD L2s (Infinity-Cache hits included): an upper bound of HBM bytes; the maps and bit planes are cache resident",
    }
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    tr = json.load(open(tp)) if os.path.exists(tp) else {}
    import subprocess
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = "unknown"
    tr[cfg] = {"hbm_bytes_per_launch": fetch_b + write_b, "round": rnd, "kernel": plan_row["Name"][:60], "commit": commit,
               "kernel_sources_sha16": bench["roofline"].get("kernel_sources_sha16"),  # (of the tree the counters were taken on: bench.py's traffic_stale)
               "file": f"profiles/{rnd}_{cfg}_counters.json"}
    json.dump(tr, open(tp, "w"), indent=1)
    # The bench line of this same collect.sh call was printed BEFORE pmc_traffic.json existed for its tree, so it carried the previous
    # round's counters marked traffic_stale.  Counters and line come from one tree and one box: the committed line
    # states the traffic measured beside it.
    rf = bench["roofline"]
    rf["traffic"] = fetch_b + write_b
    rf["traffic_source"] = {"file": f"profiles/{rnd}_{cfg}_counters.json", "round": rnd, "commit": commit, "kernel": plan_row["Name"][:60],
                            "kernel_sources_sha16": rf.get("kernel_sources_sha16"), "traffic_stale": False,
                            "note": "measured by the rocprofv3 --pmc passes of the same profiles/collect.sh call as this line (same tree, same box); "
                                    "filled in by profiles/summarise.py after the run"}
    if rf.get("kernel_ms"):
        rf["frac_by_counter_bytes"] = (fetch_b + write_b) / (rf["kernel_ms"] * 1e-3) / 1e9 / rf["peak"]
waves = allc.get("SQ_WAVES")
if waves:
    pw = {k: v / waves for k, v in sorted(allc.items()) if k.startswith("SQ_")}
    out["sq_per_wavefront"] = {"unit": "counter / SQ_WAVES per launch; *_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* are quad-cycles",
                               "waves_per_launch": waves, "per_wave": pw}
    if "SQ_ACTIVE_INST_VALU" in pw and "SQ_WAVE_CYCLES" in pw:
        out["sq_per_wavefront"]["derived"] = {"valu_active_fraction_of_wave_cycles": pw["SQ_ACTIVE_INST_VALU"] / pw["SQ_WAVE_CYCLES"],
                                              "wait_fraction_of_wave_cycles": pw.get("SQ_WAIT_ANY", 0) / pw["SQ_WAVE_CYCLES"]}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_{cfg}_counters.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(ROOT, "profiles", f"{rnd}_{cfg}_bench_line.json"), "w"), indent=1)
print(cfg, "kernel", plan_row["Name"][:50], "avg ns", plan_row["AverageNs"], "calls", plan_row["Calls"], "bench kernel_ms", bench["roofline"]["kernel_ms"],
      "frac", round(bench["roofline"]["frac"], 4), "traffic/alg", round(out.get("traffic", {}).get("traffic_over_algorithmic", 0), 3))
