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
