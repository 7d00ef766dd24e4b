#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-r4_kp}; mkdir -p $out
export TMPDIR=/tmp
BIG_MODES=6,7 BIG_MIN_TILES=1 timeout 300 python tools/big_probe.py > $out/big_probe.txt 2>&1; echo "probe rc=$?"; tail -20 $out/big_probe.txt
CONV_MODES=6,7 timeout 200 python tools/conv_probe.py > $out/conv_probe.txt 2>&1; echo "conv rc=$?"; tail -7 $out/conv_probe.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "big_tile or grouped" > $out/kernels.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/kernels.txt
