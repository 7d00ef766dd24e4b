#!/bin/bash
out=gpurun_out/r2_run31; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "gemm" 2>&1 | tail -6
PH_GEMM_STREAMK=1 PH_GEMM_BIG_GROUPED=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "streamk" 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
PH_GEMM_BIG_GROUPED=0 timeout 300 $B > $out/ab_nogrp.json 2> $out/ab_nogrp.err
timeout 300 $B > $out/ab_grp.json 2> $out/ab_grp.err
PH_GEMM_BIG_GROUPED=0 timeout 300 $B > $out/ab_nogrp2.json 2> $out/ab_nogrp2.err
timeout 300 $B > $out/ab_grp2.json 2> $out/ab_grp2.err
for f in nogrp grp nogrp2 grp2; do python - <<PY
import json
try:
    d = json.loads(open('$out/ab_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['config'].get('final_loss'))
except Exception as e:
    print('$f', 'ERR', e, open('$out/ab_$f.err').read()[-400:])
PY
done
