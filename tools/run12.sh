#!/bin/bash
out=gpurun_out/r2_run12; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q --timeout=900 -x -k "implicit or (golden and (tiny or base_b8)) or equals_eager" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $out/pytest.log | tail -12
ab() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/ab_$name.json 2> $out/ab_$name.err; python - <<PY
import json
try:
    d = json.load(open('$out/ab_$name.json')); print('$name', d['ms_per_step'], d['value'], d['config']['final_loss'])
except Exception as e:
    print('$name FAILED', e); print(open('$out/ab_$name.err').read()[-1500:])
PY
}
ab grouped A=1
ab explicit PRISMER_STEMS=explicit
ab grouped2 A=1
ab explicit2 PRISMER_STEMS=explicit
PH_PROF_DUMP=$out/grouped.csv timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $out/b_grouped.json 2> $out/b_grouped.err
python tools/percall.py $out/grouped.csv 5 70 > $out/grouped_percall.txt
