#!/bin/bash
# round 4, GPU call 1: where do the GEMM / attention / LayerNorm launches spend their time (s_memtime timeline build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4_c1
export TMPDIR=/tmp
PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_tl.so timeout 600 python tools/timeline_probe.py > gpurun_out/r4_c1/timeline.txt 2>&1
echo "timeline rc=$?"
tail -5 gpurun_out/r4_c1/timeline.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r4_c1/kernels_gemm.txt 2>&1
echo "pytest rc=$?"
tail -3 gpurun_out/r4_c1/kernels_gemm.txt
