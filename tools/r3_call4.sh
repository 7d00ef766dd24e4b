#!/bin/bash
out=gpurun_out/r3_c4; mkdir -p $out
export TMPDIR=/tmp
for mode in 1 0; do
cd /tmp; PRISMER_IMPLICIT_DGRAD=$mode timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/kt$mode -o kt -- python $OLDPWD/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $OLDPWD/$out/kt$mode.log 2>&1; cd $OLDPWD
KT=$(find $out/kt$mode -name "*.db" | head -1)
python tools/rocprof_summary.py $KT $out/kernel_stats$mode.csv 8 400 > $out/kernel_summary$mode.txt 2>&1; head -3 $out/kernel_summary$mode.txt
find $out -name "*.db" -size +20M -delete
done
