#!/bin/bash
out=gpurun_out/r2_run11; mkdir -p $out
PH_PROF_DUMP=$out/grouped.csv timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $out/b_grouped.json 2> $out/b_grouped.err
PRISMER_STEMS=explicit PH_PROF_DUMP=$out/explicit.csv timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $out/b_explicit.json 2> $out/b_explicit.err
python tools/percall.py $out/grouped.csv 5 70 > $out/grouped_percall.txt; python tools/percall.py $out/explicit.csv 5 70 > $out/explicit_percall.txt
head -45 $out/grouped_percall.txt
