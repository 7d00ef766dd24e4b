#!/bin/bash
out=gpurun_out/r3_c15; mkdir -p $out
for pre in 0 1; do
PH_GEMM_EPI_PRE=$pre BIG_MODES=5 BIG_MIN_TILES=128 timeout 300 python tools/big_probe.py > $out/big_probe_pre$pre.txt 2>&1; grep -E "vit out|vit proj|dgrad 8320x768x768|dgrad fc|dgrad qkv|vit fc|dgrad proj|ragged" $out/big_probe_pre$pre.txt
done
for cfg in "PH_GEMM_EPI_PRE=0" "PH_GEMM_EPI_PRE=1" "PH_GEMM_EPI_PRE=0" "PH_GEMM_EPI_PRE=1"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
