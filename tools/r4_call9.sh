#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_c9; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv" > $out/kernels_gemm.txt 2>&1; echo "pytest rc=$?"; tail -2 $out/kernels_gemm.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/f -o pmc -- python tools/fetch_probe.py run > $out/f.log 2>&1; echo rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/w -o pmc -- python tools/fetch_probe.py run > $out/w.log 2>&1; echo rc=$?
F=$(find $out/f -name "*.db" | head -1); W=$(find $out/w -name "*.db" | head -1)
python tools/fetch_probe.py table $F $W > $out/fetch_table.txt 2>&1; head -12 $out/fetch_table.txt | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), {k:(v.get('value') if isinstance(v,dict) else v) for k,v in (d.get('secondary') or {}).items()})"
find $out -name "*.db" -size +20M -delete
