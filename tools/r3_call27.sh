out=gpurun_out/r3_c27; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q --timeout=600 -k "wgrad or deferred or splitk or conv or grouped" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q --timeout=600 -k "golden and (base_b8 or tiny or zbase)" 2>&1 | tail -3
for cfg in "PRISMER_GROUP_WGRAD=0" "PRISMER_GROUP_WGRAD=1" "PRISMER_GROUP_WGRAD=0" "PRISMER_GROUP_WGRAD=1"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
