#!/bin/bash
out=gpurun_out/r2_run6; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest.log | tail -20
ab() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/ab_$name.json 2> $out/ab_$name.err; python - <<PY
import json
try:
    d = json.load(open('$out/ab_$name.json')); print('$name', d['ms_per_step'], d['value'], d['config']['final_loss'])
except Exception as e:
    print('$name FAILED', e)
PY
}
ab serial A=1
ab branches PRISMER_SIDE_STREAM=1 PRISMER_EXPERIMENTAL_GRAPH_BRANCHES=1
ab serial2 A=1
