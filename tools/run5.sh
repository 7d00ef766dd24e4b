#!/bin/bash
out=gpurun_out/r2_run5; mkdir -p $out
timeout 600 python tests/tools/diag_graph_grads.py tiny_caption > $out/diag_tiny.txt 2>&1; tail -40 $out/diag_tiny.txt
