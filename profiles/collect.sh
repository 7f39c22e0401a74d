#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's roofline fields on the GPU box.
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r1c'
# Raw output -> gpurun_out/<tag>_*/ ; profiles/summarise.py turns it into the committed summaries.
# Counters are collected in their own runs (--kernel-trace + --pmc only), a few per pass.
set -u
TAG=${1:-r1c}
OUT=gpurun_out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o $TAG -- $BENCH > $OUT/${TAG}_stats.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${TAG}_pmc$i -o $TAG -- $BENCH > $OUT/${TAG}_pmc$i.log 2>&1
done
# calibration of FETCH_SIZE / WRITE_SIZE on a kernel with known traffic (one 4000x4000 f32 layer transposed)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_calib_fetch -o $TAG -- python3 profiles/calib.py > $OUT/${TAG}_calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_calib_write -o $TAG -- python3 profiles/calib.py > $OUT/${TAG}_calib_write.log 2>&1
$BENCH > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
ls $OUT | grep $TAG
