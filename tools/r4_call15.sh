#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c15; mkdir -p $out
export TMPDIR=/tmp
PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_tl.so TL_SMALL=0 TL_BIG_MODES=6,7 timeout 300 python tools/timeline_probe.py > $out/timeline.txt 2>&1
echo "rc=$?"; grep -- "-- chain" $out/timeline.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv" > $out/kernels_gemm.txt 2>&1
echo "pytest rc=$?"; tail -3 $out/kernels_gemm.txt
