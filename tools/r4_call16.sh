#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c16; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv" > $out/kernels_conv.txt 2>&1
echo "pytest rc=$?"; tail -15 $out/kernels_conv.txt
timeout 300 python tools/conv_probe.py > $out/conv_probe.txt 2>&1
echo "probe rc=$?"; cat $out/conv_probe.txt | tail -12
