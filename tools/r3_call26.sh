out=gpurun_out/r3_c26; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=600 -k "deferred or splitk or conv" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q --timeout=600 -k "base_b8 or tiny" 2>&1 | tail -3
for cfg in "PRISMER_DEFER_REDUCE=0" "PRISMER_DEFER_REDUCE=1" "PRISMER_DEFER_REDUCE=0" "PRISMER_DEFER_REDUCE=1"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
