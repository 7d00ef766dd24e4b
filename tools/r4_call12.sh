#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c12; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -s -m gpu -k "test_trainer_hipgraph_step_matches_reference_golden and base_b32" > $out/parity_bf16stats.txt 2>&1; echo "rc=$?"; grep "stems:" $out/parity_bf16stats.txt | cut -c1-700
PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_csf32.so timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -s -m gpu -k "test_trainer_hipgraph_step_matches_reference_golden and base_b32" > $out/parity_fp32stats.txt 2>&1; echo "rc=$?"; grep "stems:" $out/parity_fp32stats.txt | cut -c1-700
