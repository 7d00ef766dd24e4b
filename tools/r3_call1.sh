#!/bin/bash
# round-3 GPU call 1: persistent big GEMM probe, kernel unit tests, bench + kernel trace
out=gpurun_out/r3_c1; mkdir -p $out
export TMPDIR=/tmp
BIG_MODES=0,5,7 timeout 400 python tools/big_probe.py > $out/big_probe.txt 2>&1; tail -25 $out/big_probe.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=300 > $out/pytest_kernels.log 2>&1; tail -5 $out/pytest_kernels.log
timeout 300 python bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 1500 $out/bench_n1.json; tail -3 $out/bench_n1.err
PH_GEMM_BIG=5 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $out/bench_mode5.json 2> $out/bench_mode5.err; python -c "
import json; d=json.load(open('$out/bench_mode5.json')); print('mode5', d['value'], d['ms_per_step'])"
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/kt -o kt -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $OLDPWD/$out/kt.log 2>&1; cd $OLDPWD
KT=$(find $out/kt -name "*.db" | head -1)
python tools/rocprof_summary.py $KT $out/kernel_stats.csv 14 400 > $out/kernel_summary.txt 2>&1; head -45 $out/kernel_summary.txt
find $out -name "*.db" -size +20M -delete
