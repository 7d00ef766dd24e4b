#!/bin/bash
# same-box A/B under the bench: small-query attention kernels (default) vs the streaming kernels for every launch (ph_attention_tuning(0)), interleaved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
out=gpurun_out/ab_attn_small; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2; do
  for w in 0 1; do
    timeout 300 python -c "
import sys
from prismer_amd import _lib
_lib.lib.ph_attention_tuning($w)
import bench
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-secondary', '--no-cpu-baseline']
bench.main()" > $out/bench_${w}_$rep.json 2> $out/bench_${w}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${w}_$rep.json')); f=d['kernel_families_ms_per_step']; print('small-query kernels = $w, rep $rep:', d['value'], 'images/s', d['ms_per_step'], 'ms | attention fwd', f['attention_fwd'], 'bwd', f['attention_bwd'], '| loss', d['config']['final_loss'])"
  done
done
