for cfg in "PH_GEMM_TT_RING=0" "PH_GEMM_TT_RING=1" "PH_GEMM_TT_RING=0" "PH_GEMM_TT_RING=1"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
