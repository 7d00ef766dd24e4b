#!/bin/bash
# round 4: bench + kernel trace + GEMM kernel tests on the product library (quick check of a GEMM change)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/${1:-r4_c5}; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv" > $out/kernels_gemm.txt 2>&1; echo "pytest rc=$?"; tail -2 $out/kernels_gemm.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), {k:(v.get('value') if isinstance(v,dict) else v) for k,v in (d.get('secondary') or {}).items()}, d.get('kernel_families_ms_per_step'))"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $out/kt.log 2>&1
KT=$(find $out/kt -name "*.db" | head -1)
python tools/rocprof_summary.py $KT $out/kernel_stats.csv 14 400 > $out/kernel_summary.txt 2>&1; head -32 $out/kernel_summary.txt
find $out -name "*.db" -size +20M -delete
