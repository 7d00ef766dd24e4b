#!/bin/bash
out=gpurun_out/r2_run19; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "conv or adamw or big_tile" > $out/pytest.log 2>&1; tail -6 $out/pytest.log
ab() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/ab_$name.json 2> $out/ab_$name.err; python - <<PY
import json
try:
    d = json.load(open('$out/ab_$name.json')); print('$name', d['ms_per_step'], d['value'], d['config']['final_loss'])
except Exception as e:
    print('$name FAILED', e); print(open('$out/ab_$name.err').read()[-1500:])
PY
}
ab new A=1
ab new2 A=1
