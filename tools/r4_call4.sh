#!/bin/bash
# round 4: full GPU suite + bench + kernel trace on the product library
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c4; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), {k:(v.get('value') if isinstance(v,dict) else v) for k,v in (d.get('secondary') or {}).items()}, d.get('kernel_families_ms_per_step'))"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $out/kt.log 2>&1
KT=$(find $out/kt -name "*.db" | head -1)
python tools/rocprof_summary.py $KT $out/kernel_stats.csv 14 400 > $out/kernel_summary.txt 2>&1; head -45 $out/kernel_summary.txt
find $out -name "*.db" -size +20M -delete
