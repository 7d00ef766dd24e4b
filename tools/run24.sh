#!/bin/bash
out=gpurun_out/r2_run24; mkdir -p $out
timeout 500 python tools/big_probe.py > $out/big_probe.txt 2>&1; cat $out/big_probe.txt | tail -18
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "gemm" 2>&1 | tail -4
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
PH_GEMM_BIG=1 PH_GEMM_BIG_WIDE=0 PH_GEMM_BIG_TB=0 timeout 300 $B > $out/ab_old.json 2> $out/ab_old.err
PH_GEMM_BIG=5 PH_GEMM_BIG_WIDE=0 PH_GEMM_BIG_TB=0 timeout 300 $B > $out/ab_pp.json 2> $out/ab_pp.err
PH_GEMM_BIG=5 PH_GEMM_BIG_WIDE=1 PH_GEMM_BIG_TB=0 timeout 300 $B > $out/ab_pp_wide.json 2> $out/ab_pp_wide.err
timeout 300 $B > $out/ab_pp_wide_tb.json 2> $out/ab_pp_wide_tb.err
PH_GEMM_BIG=1 PH_GEMM_BIG_WIDE=0 PH_GEMM_BIG_TB=0 timeout 300 $B > $out/ab_old2.json 2> $out/ab_old2.err
for f in old pp pp_wide pp_wide_tb old2; do python - <<PY
import json
try:
    d = json.loads(open('$out/ab_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['config'].get('final_loss'))
except Exception as e:
    print('$f', 'ERR', e, open('$out/ab_$f.err').read()[-400:])
PY
done
