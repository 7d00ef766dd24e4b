#!/bin/bash
# same-box A/B: Philox4x32 with 10 rounds (variant build p10) vs 7 rounds (product), interleaved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/r4_philox_ab; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2; do
  for v in p10 prod; do
    lib=prismer_amd/lib/libprismer_hip.so; [ $v = p10 ] && lib=prismer_amd/lib/libprismer_hip_p10.so
    PRISMER_HIP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${v}_$rep.json')); f=d['kernel_families_ms_per_step']; print('$v rep $rep:', d['value'], 'images/s', d['ms_per_step'], 'ms | attention fwd', f['attention_fwd'], 'bwd', f['attention_bwd'], '| gemm', f['gemm'], '| layernorm', f['layernorm'], '| embed_ce', f['embed_ce'])"
  done
done
