#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-r4_c17}; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q -x --timeout=600 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; tail -2 $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac')); print(d.get('kernel_families_ms_per_step'))"
