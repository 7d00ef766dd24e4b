#!/bin/bash
out=gpurun_out/r2_run25; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $out/kt.log 2>&1
KT=$(find $out/kt -name "*.db" | head -1)
python tools/rocprof_summary.py $KT $out/kernel_stats.csv 14 400 > $out/kernel_summary.txt 2>&1; head -45 $out/kernel_summary.txt
find $out -name "*.db" -size +20M -delete
