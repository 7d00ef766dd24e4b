#!/bin/bash
# Profile set of a round (the recipe behind profiles/rN_*): default bench line, rocprofv3 kernel trace of the bench under graph replay, three
# PMC passes (eager; FETCH_SIZE, WRITE_SIZE, MFMA busy -- separate runs, kernel-trace only: the pool refuses PMC + other trace domains),
# per-kernel tables, phase times, secondary workload traces, two-rank dry run on one GPU.      bash tools/profile_round.sh r5 [outdir]
p=${1:-r5}; out=${2:-gpurun_out/${p}_prof}; mkdir -p $out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 > $out/${p}_bench_n1.json 2> $out/bench_n1.err; tail -c 300 $out/${p}_bench_n1.json; echo
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $out/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $out/pmc_$c.log 2>&1
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_MFMA -o pmc -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $out/pmc_MFMA.log 2>&1
KT=$(find $out/kt -name "*.db" | head -1); F=$(find $out/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find $out/pmc_WRITE_SIZE -name "*.db" | head -1); M=$(find $out/pmc_MFMA -name "*.db" | head -1)
python tools/rocprof_summary.py $KT $out/${p}_kernel_stats.csv 14 400 > $out/${p}_kernel_summary.txt 2>&1; head -5 $out/${p}_kernel_summary.txt
python tools/profile_tables.py $KT 14 400 $F $W $M 3 $out/${p} 2>&1 | head -30
python tools/pmc_summary.py $F $W 3 $out/${p}_pmc_gemm.json > /dev/null 2>&1
timeout 300 python tools/phase_times.py 2>&1 | grep -v amdgpu.ids > $out/${p}_phase_times.txt; cat $out/${p}_phase_times.txt
PRISMER_DIST_BACKEND=gloo PRISMER_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $out/${p}_bench_dryrun_2ranks_gloo_one_gpu.json 2> $out/dryrun_2ranks.err; tail -c 400 $out/${p}_bench_dryrun_2ranks_gloo_one_gpu.json; tail -2 $out/dryrun_2ranks.err
find $out -name "*.db" -size +20M -delete          # keep the merged output small
