#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c6; mkdir -p $out
export TMPDIR=/tmp
PH_GEMM_KS4=0 timeout 300 python tools/ks_probe.py > $out/ks2.txt 2>&1; echo rc=$?; cat $out/ks2.txt | tail -12
PH_GEMM_KS4=1 timeout 300 python tools/ks_probe.py > $out/ks4.txt 2>&1; echo rc=$?; cat $out/ks4.txt | tail -12
