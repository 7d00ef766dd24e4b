#!/bin/bash
out=gpurun_out/r2_run27; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
for cfg in new old; do
  if [ $cfg = old ]; then export PH_GEMM_BIG=1 PH_GEMM_BIG_WIDE=0 PH_GEMM_BIG_TB=0; fi
  timeout 900 rocprofv3 --kernel-trace -d $out/kt_$cfg -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $out/kt_$cfg.log 2>&1
  KT=$(find $out/kt_$cfg -name "*.db" | head -1)
  python tools/rocprof_summary.py $KT $out/kernel_stats_$cfg.csv 14 400 > $out/kernel_summary_$cfg.txt 2>&1
  grep -A 30 "by kernel and grid" $out/kernel_summary_$cfg.txt | head -34
done
find $out -name "*.db" -size +20M -delete
