#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c10; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "test_attention or huge" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; tail -3 $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac')); print(json.dumps(d.get('secondary'), indent=1)[:2500]); print(d.get('cpu_baseline'))"
