#!/bin/bash
out=gpurun_out/r2_run16; mkdir -p $out
ab() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/ab_$name.json 2> $out/ab_$name.err; python - <<PY
import json
try:
    d = json.load(open('$out/ab_$name.json')); print('$name', d['ms_per_step'], d['value'], d['config']['final_loss'])
except Exception as e:
    print('$name FAILED', e); print(open('$out/ab_$name.err').read()[-1500:])
PY
}
ab base A=1
ab adamw_streams PRISMER_ADAMW_OVERLAP=2
ab base2 A=1
ab adamw_streams2 PRISMER_ADAMW_OVERLAP=2
