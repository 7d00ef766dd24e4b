#!/bin/bash
# same-box A/B of two library builds under the bench: the product library vs prismer_amd/lib/libprismer_hip_$1.so (tools/build_variant.py), interleaved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
v=${1:-prev}; out=gpurun_out/ab_$v; mkdir -p $out; export TMPDIR=/tmp
if [ -n "$2" ]; then timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "$2" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.txt; fi
for rep in 1 2; do
  for w in $v prod; do
    lib=prismer_amd/lib/libprismer_hip.so; [ $w != prod ] && lib=prismer_amd/lib/libprismer_hip_$w.so
    PRISMER_HIP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $out/bench_${w}_$rep.json 2> $out/bench_${w}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${w}_$rep.json')); f=d['kernel_families_ms_per_step']; print('$w rep $rep:', d['value'], 'images/s', d['ms_per_step'], 'ms | attention fwd', f['attention_fwd'], 'bwd', f['attention_bwd'], '| gemm', f['gemm'], '| layernorm', f['layernorm'], '| frontend', f['frontend'], '| optimizer', f['optimizer'])"
  done
done
