#!/bin/bash
out=gpurun_out/r3_c2; mkdir -p $out
export TMPDIR=/tmp
PH_ATTN_PLAIN=0 timeout 300 python tools/attn_probe.py > $out/attn_probe_generic.txt 2>&1; tail -6 $out/attn_probe_generic.txt
PH_ATTN_PLAIN=1 timeout 300 python tools/attn_probe.py > $out/attn_probe_plain.txt 2>&1; tail -6 $out/attn_probe_plain.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=300 > $out/pytest_kernels.log 2>&1; tail -3 $out/pytest_kernels.log
for ts in 0 1; do
PH_GEMM_TAIL_SPLIT=$ts BIG_MODES=5 BIG_MIN_TILES=128 BIG_WIDE_ONLY=1 timeout 300 python tools/big_probe.py > $out/big_probe_tail$ts.txt 2>&1; tail -7 $out/big_probe_tail$ts.txt
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_families_ms_per_step'])"; tail -3 $out/bench_n1.err
PH_GEMM_TAIL_SPLIT=0 PH_ATTN_PLAIN=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $out/bench_base.json 2> $out/bench_base.err; python -c "
import json; d=json.load(open('$out/bench_base.json')); print('bench base', d['value'], d['ms_per_step'])"
