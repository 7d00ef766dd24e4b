#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c11; mkdir -p $out
export TMPDIR=/tmp
( time timeout 150 python -c "import bench, json; print(json.dumps(bench.dropin_leg(steps=5, warmup=2)))" ) > $out/dropin.txt 2>&1; echo "dropin rc=$?"; tail -4 $out/dropin.txt | cut -c1-600
( time timeout 150 python -c "import bench, json; print(json.dumps(bench.loader_leg(steps=5, warmup=2)))" ) > $out/loader.txt 2>&1; echo "loader rc=$?"; tail -4 $out/loader.txt | cut -c1-600
( time timeout 150 python -c "import bench, json; print(json.dumps(bench.cpu_baseline()))" ) > $out/cpu.txt 2>&1; echo "cpu rc=$?"; tail -4 $out/cpu.txt | cut -c1-900
