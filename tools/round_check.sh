#!/bin/bash
# End-of-round check on the GPU box: the whole -m gpu suite, smoke(), the default bench line.      bash tools/round_check.sh [name]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-r5_final}; mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --timeout=1200 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"; tail -2 $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('dominant_kernel', {}).get('frac'), d['roofline']['traffic'], d['roofline']['traffic_source']); print({k:(v.get('value'), v.get('ms_per_step'), v.get('final_loss'), v.get('error')) for k,v in (d.get('secondary') or {}).items()}); print(d.get('cpu_baseline'))"
