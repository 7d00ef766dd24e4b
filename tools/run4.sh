#!/bin/bash
out=gpurun_out/r2_run4; mkdir -p $out
timeout 600 python tests/tools/diag_graph_grads.py tiny_caption > $out/diag_tiny.txt 2>&1; tail -12 $out/diag_tiny.txt
timeout 600 python tests/tools/diag_graph_grads.py zbase_b4 > $out/diag_zbase.txt 2>&1; tail -12 $out/diag_zbase.txt
timeout 300 python -m pytest tests -m gpu -q --durations=12 -k "not hipgraph_step and not equals_eager" --timeout=600 2>&1 | tail -25
