#!/bin/bash
out=gpurun_out/r3_c6; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  |rc=" $out/pytest.log | tail -12
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 2500 $out/bench_n1.json; tail -3 $out/bench_n1.err
