#!/bin/bash
# kernel traces of the secondary workloads (BASELINE configs 2 and 5) under graph replay
p=${1:-r5}; out=${2:-gpurun_out/${p}_sec}; mkdir -p $out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo; export TMPDIR=/tmp
for w in "large_vqa 16" "z_base_caption 32"; do
  set -- $w
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$1 -o kt -- python bench.py --workload $1 --batch $2 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $out/kt_$1.log 2>&1
  db=$(find $out/kt_$1 -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $out/${p}_kernel_stats_$1.csv 8 2000 > $out/${p}_kernel_summary_$1.txt 2>&1; head -12 $out/${p}_kernel_summary_$1.txt
  tail -c 600 $out/kt_$1.log | grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": 12, "warmup": 3, "ms_per_step": [0-9.]*'
done
find $out -name "*.db" -delete
