#!/bin/bash
# GPU call A of round 2: the whole -m gpu suite (with durations), a bench line per configuration with the direct
# kernels, the 2-rank launcher path on one GPU, and rocprofv3 kernel stats per configuration.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2a_pytest.log
tail -25 gpurun_out/r2a_pytest.log
python3 bench.py > gpurun_out/r2a_bench_headline.json 2> gpurun_out/r2a_bench_headline.err; echo "bench headline rc=$?"
for c in cfg2 cfg3 cfg4 cfg5; do
  python3 bench.py --config $c --steps 20 --no-cpu-baseline --no-extras > gpurun_out/r2a_bench_$c.json 2> gpurun_out/r2a_bench_$c.err; echo "bench $c rc=$?"
done
FPE_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2a_bench_2rank_shared.json 2> gpurun_out/r2a_bench_2rank_shared.err; echo "2-rank rc=$?"
for c in headline cfg3 cfg4 cfg5; do
  bash profiles/collect.sh r2a_$c $c stats > /dev/null 2>&1
done
cat gpurun_out/r2a_bench_headline.json | head -c 3000
