#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's roofline fields on the GPU box.
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh <tag> [config] [full|lite|stats] [extra bench args]'
# Raw output -> gpurun_out/<tag>_*/ ; profiles/summarise.py turns it into the committed summaries.
# Counters are collected in their own runs (--kernel-trace + --pmc only), a few per pass; the program follows
# `--` directly (python3 bench.py ...), never a shell or env wrapper.
#   stats: kernel trace + stats only;  lite: + FETCH/WRITE + instruction/cycle counters;  full: every group + calibration
set -u
TAG=${1:-r2}
CFG=${2:-headline}
MODE=${3:-full}
EXTRA=${4:-}
OUT=gpurun_out
export TMPDIR=/tmp
STEPS=50
case $CFG in cfg3|cfg4|cfg5) STEPS=20;; esac
# (--b2b-seconds 0: the untimed back-to-back pass stays at its minimum of 200 launches — the profiler's per-kernel AVERAGE must describe the
#  timed region's launches, which start from an idle GPU block by block and run ~7 % longer than launches that keep the queue busy)
BENCH="python3 bench.py --config $CFG --steps $STEPS --warmup 5 --blocks 3 --b2b-seconds 0 --no-cpu-baseline --no-extras $EXTRA"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o $TAG -- $BENCH > $OUT/${TAG}_stats.log 2>&1
if [ "$MODE" != "stats" ]; then
  if [ "$MODE" = "full" ]; then
    GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
             "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
             "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
             "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH")
  else
    GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY")
  fi
  i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${TAG}_pmc$i -o $TAG -- $BENCH > $OUT/${TAG}_pmc$i.log 2>&1
  done
  if [ "$MODE" = "full" ]; then
    # calibration of FETCH_SIZE / WRITE_SIZE on a kernel with known traffic (one 4000x4000 f32 layer transposed)
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_calib_fetch -o $TAG -- python3 profiles/calib.py > $OUT/${TAG}_calib_fetch.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_calib_write -o $TAG -- python3 profiles/calib.py > $OUT/${TAG}_calib_write.log 2>&1
  fi
fi
$BENCH > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
ls $OUT | grep "^${TAG}_" | head -40
