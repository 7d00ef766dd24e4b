out=gpurun_out/r3_c19; mkdir -p $out
for v in "" pad16; do
  if [ -n "$v" ]; then export PRISMER_HIP_LIB=$PWD/prismer_amd/lib/libprismer_hip_$v.so; fi
  echo "== variant: ${v:-product}"
  timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out/attn_probe.txt
done
unset PRISMER_HIP_LIB
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q --timeout=600 -k "large_vqa_b1" 2>&1 | tail -3
