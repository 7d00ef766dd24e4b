out=gpurun_out/r3_c17; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=300 -k "attention or attn" > $out/pytest_attn.log 2>&1; tail -4 $out/pytest_attn.log
timeout 300 python tools/attn_probe.py > $out/attn_probe.txt 2>&1; cat $out/attn_probe.txt
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['config']['final_loss'], d.get('kernel_families_ms_per_step'))"
done
