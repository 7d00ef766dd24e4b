#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-r4_c13}; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; tail -2 $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac')); print({k:(v.get('value'), v.get('ms_per_step')) if isinstance(v,dict) else v for k,v in (d.get('secondary') or {}).items()}); print(d.get('cpu_baseline')); print(d.get('kernel_families_ms_per_step'))"
