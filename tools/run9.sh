#!/bin/bash
out=gpurun_out/r2_run9; mkdir -p $out
timeout 600 python tests/tools/diag_rng.py > $out/diag_rng.txt 2>&1; tail -8 $out/diag_rng.txt
