out=gpurun_out/r3_c21; mkdir -p $out
for v in "" dq4 dkv4 bwd4; do
  if [ -n "$v" ]; then export PRISMER_HIP_LIB=$PWD/prismer_amd/lib/libprismer_hip_$v.so; fi
  echo "== variant: ${v:-product}"
  timeout 300 python tools/attn_probe.py 2>&1 | grep -v "amdgpu.ids\|PH_ATTN" | tee -a $out/attn_probe.txt
done
