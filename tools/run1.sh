#!/bin/bash
# GPU call 1 of round 2: full GPU test suite, the default bench line, then A/B of the scheduling switches.
out=gpurun_out/r2_run1; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json
ab() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/ab_$name.json 2> $out/ab_$name.err; python - <<PY
import json
try:
    d = json.load(open('$out/ab_$name.json')); print('$name', d['ms_per_step'], d['value'])
except Exception as e:
    print('$name FAILED', e)
PY
}
ab base A=1
ab noadamw PRISMER_ADAMW_OVERLAP=0
ab nosavegrad PRISMER_SAVE_ACT_GRAD=0
ab bg64 PRISMER_WGRAD_EAGER_FLUSH=1 PRISMER_WGRAD_BG_BLOCKS=64
ab bg128 PRISMER_WGRAD_EAGER_FLUSH=1 PRISMER_WGRAD_BG_BLOCKS=128
ab bg256 PRISMER_WGRAD_EAGER_FLUSH=1 PRISMER_WGRAD_BG_BLOCKS=256
ab base2 A=1
