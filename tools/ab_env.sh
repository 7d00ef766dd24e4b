#!/bin/bash
# A/B of environment switches on one box: tools/ab_env.sh <outdir> "<bench args>" "<label>=<ENV=.. ENV=..>" ...   (label=- for no env)
out=$1; shift; args=$1; shift; mkdir -p $out
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}; [ "$envs" = "-" ] && envs=""
  env $envs timeout 400 python bench.py --no-cpu-baseline --no-roofline $args > $out/ab_$label.json 2> $out/ab_$label.err
  python - <<PY
import json
try:
    d = json.loads(open('$out/ab_$label.json').read().strip().splitlines()[-1]); print('%-16s %8.3f ms  %9.2f  loss %s   [$envs]' % ('$label', d['ms_per_step'], d['value'], d['config'].get('final_loss')))
except Exception as e:
    print('$label', 'ERR', e, open('$out/ab_$label.err').read()[-400:])
PY
done
