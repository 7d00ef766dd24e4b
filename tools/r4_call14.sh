#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c14; mkdir -p $out
export TMPDIR=/tmp
PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_wsnl.so TL_SMALL=0 TL_BIG_MODES=7 timeout 300 python tools/timeline_probe.py > $out/timeline_wsnl.txt 2>&1
echo "rc=$?"; grep -- "-- chain" $out/timeline_wsnl.txt; grep -A1 "m7 big proj 8320x768x3072 +res \[warm\]" $out/timeline_wsnl.txt | cut -c1-250
