#!/bin/bash
# profiles/collect_filters_counters.sh MAPIDX LIB...: instruction counters per kernel of the filter chain on one map, per variant library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FPE_PROBE_MAP=$1; shift
mkdir -p gpurun_out/r5pmc
for v in "$@"; do
  export FPE_LIB=$GRAFT_REPO_ROOT/scratch/libfpe_$v.so
  [ "$v" = "tree" ] && unset FPE_LIB
  rm -rf gpurun_out/r5pmc/$v
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_CVT"; do
    d=gpurun_out/r5pmc/$v/$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o f -- python3 profiles/probe_filters.py > /dev/null 2>&1
  done
  python3 - "$v" "$FPE_PROBE_MAP" <<'PY'
import collections, csv, glob, sys
v = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/r5pmc/{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"== {v} (map {sys.argv[2]})")
for k, c in cnt.items():
    m = {n: sum(x) / len(x) for n, x in c.items()}
    w = m["SQ_WAVES"]
    print(f"{k}: waves {w:.0f} VALU {m['SQ_INSTS_VALU']/w:.0f} SALU {m['SQ_INSTS_SALU']/w:.0f} LDS {m['SQ_INSTS_LDS']/w:.0f} | f64 add {m['SQ_INSTS_VALU_ADD_F64']/w:.0f} mul {m['SQ_INSTS_VALU_MUL_F64']/w:.0f} fma {m['SQ_INSTS_VALU_FMA_F64']/w:.0f} trans {m['SQ_INSTS_VALU_TRANS_F64']/w:.0f} cvt {m['SQ_INSTS_VALU_CVT']/w:.0f} int {m['SQ_INSTS_VALU_INT32']/w:.0f} | life {4*m['SQ_WAVE_CYCLES']/w:.0f} clk valu-active {4*m['SQ_ACTIVE_INST_VALU']/w:.0f}")
PY
done
