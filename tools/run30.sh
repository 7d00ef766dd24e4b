#!/bin/bash
out=gpurun_out/r2_run30; mkdir -p $out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
PH_GEMM_STREAMK=0 timeout 300 $B > $out/ab_nosk.json 2> $out/ab_nosk.err
timeout 300 $B > $out/ab_sk64.json 2> $out/ab_sk64.err
PH_GEMM_STREAMK_MIN_KT=200 timeout 300 $B > $out/ab_sk200.json 2> $out/ab_sk200.err
PH_GEMM_STREAMK=0 timeout 300 $B > $out/ab_nosk2.json 2> $out/ab_nosk2.err
for f in nosk sk64 sk200 nosk2; do python - <<PY
import json
try:
    d = json.loads(open('$out/ab_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['config'].get('final_loss'))
except Exception as e:
    print('$f', 'ERR', e, open('$out/ab_$f.err').read()[-400:])
PY
done
