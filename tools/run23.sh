#!/bin/bash
out=gpurun_out/r2_run23; mkdir -p $out
BIG_MODES=0,1,5,7 timeout 420 python tools/big_probe.py > $out/big_probe.txt 2>&1; cat $out/big_probe.txt | tail -12
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "gemm" 2>&1 | tail -4
