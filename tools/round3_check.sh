#!/bin/bash
# end-of-round verification on the GPU box: full -m gpu suite (with the printed parity figures), smoke, the profile set (tools/profile_round3.sh)
out=gpurun_out/r3_check; mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -s --timeout=900 --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  |rc=" $out/pytest.log | tail -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
bash tools/profile_round3.sh gpurun_out/r3_prof > $out/profile.log 2>&1; grep -E "per step|total " $out/profile.log | head -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --compact-labels > gpurun_out/r3_prof/bench_compact_labels.json 2>/dev/null
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary --freeze none > gpurun_out/r3_prof/bench_freeze_none.json 2>/dev/null
for f in bench_n1 bench_compact_labels bench_freeze_none; do python -c "
import json; d=json.load(open('gpurun_out/r3_prof/$f.json')); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('secondary'))"; done
