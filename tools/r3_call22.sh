for cfg in "PRISMER_EXP_OVERWRITE=0" "PRISMER_EXP_OVERWRITE=1" "PRISMER_EXP_OVERWRITE=0" "PRISMER_EXP_OVERWRITE=1"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
