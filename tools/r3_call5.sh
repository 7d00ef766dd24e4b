#!/bin/bash
out=gpurun_out/r3_c5; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=300 -k "conv" > $out/pytest_kernels.log 2>&1; tail -4 $out/pytest_kernels.log
for ring in 1 0; do
cd /tmp; PH_GEMM_CONV_RING=$ring timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/kt$ring -o kt -- python $OLDPWD/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $OLDPWD/$out/kt$ring.log 2>&1; cd $OLDPWD
KT=$(find $out/kt$ring -name "*.db" | head -1)
python tools/rocprof_summary.py $KT $out/kernel_stats$ring.csv 8 400 > $out/kernel_summary$ring.txt 2>&1; head -1 $out/kernel_summary$ring.txt
find $out -name "*.db" -size +20M -delete
done
