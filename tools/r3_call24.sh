out=gpurun_out/r3_c24; mkdir -p $out
for i in 1; do
timeout 1500 python -X faulthandler -m pytest tests/test_parity_gpu.py tests/test_dp_gpu.py tests/test_kernels_gpu.py -x -q --timeout=900 -k "trainer or dp or sharded or reduce_scatter or adamw or resume or optimizer" > $out/pytest$i.log 2>&1; tail -3 $out/pytest$i.log
done
