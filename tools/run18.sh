#!/bin/bash
out=gpurun_out/r2_run18; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $out/pytest.log | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
bash tools/profile_round.sh gpurun_out/r2_prof > $out/profile.log 2>&1; tail -45 $out/profile.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --compact-labels > gpurun_out/r2_prof/bench_compact_labels.json 2>/dev/null; tail -c 300 gpurun_out/r2_prof/bench_compact_labels.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --workload z_base_caption > gpurun_out/r2_prof/bench_prismerz_base.json 2>/dev/null; tail -c 200 gpurun_out/r2_prof/bench_prismerz_base.json
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --workload large_vqa --batch 16 > gpurun_out/r2_prof/bench_large_vqa_bs16.json 2>/dev/null; tail -c 200 gpurun_out/r2_prof/bench_large_vqa_bs16.json
