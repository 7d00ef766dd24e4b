out=gpurun_out/r3_c20; mkdir -p $out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_dp_gpu.py -x -q --timeout=600 -k "graph_equals_eager or dp or resume or sharded or reduce_scatter or base_b8" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
