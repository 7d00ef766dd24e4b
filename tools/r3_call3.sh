#!/bin/bash
out=gpurun_out/r3_c3; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=300 -k "conv or gemm" > $out/pytest_kernels.log 2>&1; tail -15 $out/pytest_kernels.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_families_ms_per_step'], d['config']['final_loss'])"; tail -3 $out/bench_n1.err
PRISMER_IMPLICIT_DGRAD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $out/bench_base.json 2> $out/bench_base.err; python -c "
import json; d=json.load(open('$out/bench_base.json')); print('bench dcol path', d['value'], d['ms_per_step'], d['config']['final_loss'])"
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q --timeout=600 -k "trainer_hipgraph or train_mode" > $out/pytest_parity.log 2>&1; tail -15 $out/pytest_parity.log
