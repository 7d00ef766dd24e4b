out=gpurun_out/r3_c23; mkdir -p $out
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_dp_gpu.py tests/test_kernels_gpu.py -x -q --timeout=900 -k "trainer or dp or sharded or reduce_scatter or adamw or resume or optimizer" > $out/pytest.log 2>&1; tail -4 $out/pytest.log
for cfg in "PRISMER_WGRAD_OVERWRITE=0" "PRISMER_WGRAD_OVERWRITE=1" "PRISMER_WGRAD_OVERWRITE=0" "PRISMER_WGRAD_OVERWRITE=1"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
