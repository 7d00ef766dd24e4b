#!/bin/bash
out=gpurun_out/r2_run29; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "gemm" 2>&1 | tail -6
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
PH_GEMM_STREAMK=0 timeout 300 $B > $out/ab_nosk.json 2> $out/ab_nosk.err
timeout 300 $B > $out/ab_sk.json 2> $out/ab_sk.err
PH_GEMM_STREAMK=0 PH_GEMM_BIG=1 PH_GEMM_BIG_TB=0 timeout 300 $B > $out/ab_old.json 2> $out/ab_old.err
timeout 300 $B > $out/ab_sk2.json 2> $out/ab_sk2.err
for f in nosk sk old sk2; do python - <<PY
import json
try:
    d = json.loads(open('$out/ab_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['config'].get('final_loss'))
except Exception as e:
    print('$f', 'ERR', e, open('$out/ab_$f.err').read()[-400:])
PY
done
