#!/bin/bash
out=gpurun_out/r3_c10; mkdir -p $out
for cfg in "PH_GEMM_BIG_WIDE=0" "PH_GEMM_BIG_WIDE=1" "PH_GEMM_BIG_WIDE=0" "PH_GEMM_BIG_WIDE=1"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
