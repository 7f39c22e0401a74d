"""gpurun_out/prof_filters/f_kernel_trace.csv + gpurun_out/probe_filters.txt -> profiles/round2_filters.txt"""
import collections, csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_filters/**/*kernel_trace.csv", recursive=True)[0])))
rows = [r for r in rows if "filter_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# launches come in chains of three (normals [+ roughness], step heights, step + sum), seven chains per map (2 warm-up + 5 timed), three maps
names = ["1000x1000 @ 0.02 m", "2000x2000 @ 0.01 m", "2000x2000 @ 0.005 m"]
per = len(rows) // 3
print("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 profiles/probe_filters.py   (MI355X; default filter chain;")
print("# the roughness filter rides in the normals kernel because both radii are 0.05 m)")
for line in open("gpurun_out/probe_filters.txt"):
    if " m: " in line: print(line.rstrip())
print()
for m, name in enumerate(names):
    acc = collections.defaultdict(list)
    for r in rows[m * per:(m + 1) * per]:
        acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(name + ": " + ", ".join(f"{k} {sum(v) / len(v):.1f} us x{len(v)}" for k, v in acc.items()))

# counters (one rocprofv3 --pmc pass per group), per launch and kernel, the 2000x2000 @ 0.01 m map (launches 8..14 of each kernel)
import json
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_filters_pmc_*/**/*counter_collection.csv", recursive=True):
    # the launches of one file in dispatch order: chains of three kernels, seven chains per map, three maps — the second map's
    # launches are the ones summarised (kernel names carry the tile edge and may differ between the maps)
    rows_f = [r for r in csv.DictReader(open(f)) if "filter_" in r["Kernel_Name"]]
    disp = sorted({int(r["Dispatch_Id"]) for r in rows_f})
    third = len(disp) // 3
    middle = set(disp[third:2 * third])
    for r in rows_f:
        if int(r["Dispatch_Id"]) in middle:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
if cnt:
    cal = json.load(open("profiles/round4_headline_counters.json"))["calibration"]
    print("\ncounters per launch, 2000x2000 @ 0.01 m (mean over the map's launches):")
    for k, c in cnt.items():
        m = {name: sum(v) / len(v) for name, v in c.items() if v}
        w = m.get("SQ_WAVES", 0.0)
        line = f"{k}: waves {w:.0f}"
        if w and "SQ_INSTS_VALU" in m: line += f", VALU instructions per wavefront {m['SQ_INSTS_VALU'] / w:.0f}, SALU {m.get('SQ_INSTS_SALU', 0) / w:.0f}, LDS {m.get('SQ_INSTS_LDS', 0) / w:.0f}"
        if w and "SQ_WAVE_CYCLES" in m and "SQ_ACTIVE_INST_VALU" in m:
            line += f"; wavefront lifetime {4 * m['SQ_WAVE_CYCLES'] / w:.0f} clk, VALU-active {4 * m['SQ_ACTIVE_INST_VALU'] / w:.0f} clk per wavefront"
        if w and "SQ_INSTS_VALU_ADD_F64" in m and "SQ_INSTS_VALU" in m:
            f64 = m["SQ_INSTS_VALU_ADD_F64"] + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_FMA_F64", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0)
            line += (f"; f64 arithmetic {f64 / w:.0f} per wavefront = {f64 / m['SQ_INSTS_VALU']:.0%} of the VALU instructions "
                     f"(add {m['SQ_INSTS_VALU_ADD_F64'] / w:.0f}, mul {m.get('SQ_INSTS_VALU_MUL_F64', 0) / w:.0f}, fma {m.get('SQ_INSTS_VALU_FMA_F64', 0) / w:.0f}, "
                     f"division / sqrt steps {m.get('SQ_INSTS_VALU_TRANS_F64', 0) / w:.0f}), conversions {m.get('SQ_INSTS_VALU_CVT', 0) / w:.0f}, integer arithmetic {m.get('SQ_INSTS_VALU_INT32', 0) / w:.0f}")
        if "FETCH_SIZE" in m: line += f"; fabric read {m['FETCH_SIZE'] * 1024 * cal['fetch_factor'] / 1e6:.1f} MB, written {m.get('WRITE_SIZE', 0) * 1024 * cal['write_factor'] / 1e6:.1f} MB (algorithmic: 16 MB per layer)"
        print(line)
