#!/bin/bash
out=gpurun_out/r3_c13; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=300 > $out/pytest_kernels.log 2>&1; tail -3 $out/pytest_kernels.log
timeout 300 python tools/attn_probe.py 2>&1 | tail -5
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_families_ms_per_step'])"
done
