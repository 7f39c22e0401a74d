#!/bin/bash
# GPU call C: parity suite + bench lines of the large-window configurations with and without the bit-window kernels.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r2c_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2c_pytest.log
tail -30 gpurun_out/r2c_pytest.log
for c in ${CONFIGS:-cfg3 cfg5 cfg4 headline}; do
  python3 bench.py --config $c --steps 20 --no-cpu-baseline --no-extras > gpurun_out/r2c_bench_$c.json 2> gpurun_out/r2c_bench_$c.err; echo "bench $c rc=$?"
  python3 bench.py --config $c --steps 20 --no-cpu-baseline --no-extras --no-bits > gpurun_out/r2c_bench_${c}_nobits.json 2> gpurun_out/r2c_bench_${c}_nobits.err; echo "bench $c nobits rc=$?"
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c_bench_*.json')):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('r2c_bench_')[1], 'kernel_ms %.4f'%l['roofline']['kernel_ms'], 'ms/step %.4f'%l['ms_per_step'], 'frac %.4f'%l['roofline']['frac'], 'verified', l['config']['verified'], l['config'].get('verify_error','')[:300])
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
