#!/bin/bash
# round-4 profile set: kernel trace (graph replays), three PMC passes (eager), default bench line, two-rank dry run
out=${1:-gpurun_out/r4_prof}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
R=$PWD
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 400 $out/bench_n1.json; echo
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $out/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $out/pmc_$c.log 2>&1
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_MFMA -o pmc -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $out/pmc_MFMA.log 2>&1
find $out -name "*.db" | head; find $out -name "*stats*" | head
KT=$(find $out/kt -name "*.db" | head -1); F=$(find $out/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find $out/pmc_WRITE_SIZE -name "*.db" | head -1); M=$(find $out/pmc_MFMA -name "*.db" | head -1)
python tools/rocprof_summary.py $KT $out/r4_kernel_stats.csv 14 400 > $out/r4_kernel_summary.txt 2>&1; head -5 $out/r4_kernel_summary.txt
python tools/profile_tables.py $KT 14 400 $F $W $M 3 $out/r4 2>&1 | head -30
python tools/pmc_summary.py $F $W 3 $out/r4_pmc_gemm.json > /dev/null 2>&1
PRISMER_DIST_BACKEND=gloo PRISMER_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $out/dryrun_2ranks.json 2> $out/dryrun_2ranks.err; tail -c 700 $out/dryrun_2ranks.json; tail -3 $out/dryrun_2ranks.err
# keep the merged output small
find $out -name "*.db" -size +20M -delete
