#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c8; mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/f -o pmc -- python tools/fetch_probe.py run > $out/f.log 2>&1; echo rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/w -o pmc -- python tools/fetch_probe.py run > $out/w.log 2>&1; echo rc=$?
F=$(find $out/f -name "*.db" | head -1); W=$(find $out/w -name "*.db" | head -1)
python tools/fetch_probe.py table $F $W > $out/fetch_table.txt 2>&1; cat $out/fetch_table.txt
find $out -name "*.db" -size +20M -delete
