#!/bin/bash
out=gpurun_out/r3_c11; mkdir -p $out
PH_ATTN_QT2=0 timeout 300 python tools/attn_probe.py > $out/attn_probe_qt1.txt 2>&1; tail -6 $out/attn_probe_qt1.txt
PH_ATTN_QT2=1 timeout 300 python tools/attn_probe.py > $out/attn_probe_qt2.txt 2>&1; tail -6 $out/attn_probe_qt2.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=300 -k "attention" > $out/pytest_kernels.log 2>&1; tail -3 $out/pytest_kernels.log
for cfg in "PH_ATTN_QT2=0" "PH_ATTN_QT2=1" "PH_ATTN_QT2=0" "PH_ATTN_QT2=1"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
