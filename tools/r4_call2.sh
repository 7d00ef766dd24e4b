#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4_c2
export TMPDIR=/tmp
PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_tl.so TL_SMALL=0 timeout 600 python tools/timeline_probe.py > gpurun_out/r4_c2/timeline.txt 2>&1
echo "timeline rc=$?"
grep -- "-- chain" gpurun_out/r4_c2/timeline.txt
PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_tl.so timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv" > gpurun_out/r4_c2/kernels_gemm_tl.txt 2>&1
echo "pytest(tl lib) rc=$?"; tail -3 gpurun_out/r4_c2/kernels_gemm_tl.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv" > gpurun_out/r4_c2/kernels_gemm.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r4_c2/kernels_gemm.txt
