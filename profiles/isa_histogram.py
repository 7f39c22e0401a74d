#!/usr/bin/env python3
"""Static ISA histogram per kernel of the engine (no GPU needed):
   python3 profiles/isa_histogram.py > profiles/round2_isa_histogram.txt
compiles csrc/fpe_kernels.hip with the production flags plus -save-temps in a temporary directory and counts the
instructions of every kernel by class; VGPRs, scratch and occupancy from the kernel descriptors' comments."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "quadrupedal_foothold_planner_amd", "csrc", "fpe_kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
def klass(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "valu_lane(spill)"
    if op.startswith("v_cmp"): return "valu_cmp"
    if op.startswith("v_"): return "valu_f64" if "f64" in op else "valu_other"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith(("s_load", "s_buffer")): return "smem"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"
with tempfile.TemporaryDirectory() as tmp:
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-save-temps", "-c", SRC, "-o", "k.o"], cwd=tmp, check=True, capture_output=True)
    lines = open(os.path.join(tmp, "fpe_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
print("Static ISA histogram per kernel (hipcc " + " ".join(FLAGS) + " --save-temps; profiles/isa_histogram.py);")
print("dynamic per-wavefront counts are in round2_<cfg>_counters.json (SQ_INSTS_*).\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_ZN3fpe\w+):", lines[i])
    if not m:
        i += 1
        continue
    name, cls, ops, info = m.group(1), collections.Counter(), collections.Counter(), {}
    i += 1
    while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
        t = lines[i].split(";")[0].strip()
        if t and not t.endswith(":") and not t.startswith("."):
            op = t.split()[0]
            cls[klass(op)] += 1
            ops[op] += 1
        i += 1
    while i < len(lines) and "; Occupancy" not in lines[i]:
        for key in ("NumVgprs", "ScratchSize", "NumSgprs"):
            mm = re.search(r"; %s: (\d+)" % key, lines[i])
            if mm: info[key] = int(mm.group(1))
        mm = re.search(r"sgpr_spill_count (\d+)|SGPRSpill: (\d+)|; SGPR spills?: (\d+)", lines[i])
        i += 1
    occ = re.search(r"; Occupancy: (\d+)", lines[i]).group(1) if i < len(lines) else "?"
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    print(demangled)
    print(f"  static instructions: {sum(cls.values())}  {dict(sorted(cls.items()))}")
    print(f"  VGPRs {info.get('NumVgprs')}, scratch {info.get('ScratchSize')} B/lane, occupancy {occ} waves/SIMD")
    print("  top opcodes: " + ", ".join(f"{o} {c}" for o, c in ops.most_common(14)) + "\n")
