#!/bin/bash
out=gpurun_out/r2_run15; mkdir -p $out
timeout 600 python tools/graph_overlap_probe.py 32 > $out/overlap32.txt 2>&1; tail -3 $out/overlap32.txt
timeout 600 python tools/graph_overlap_probe.py 16 > $out/overlap16.txt 2>&1; tail -3 $out/overlap16.txt
timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q --timeout=900 -k "sharded" 2>&1 | tail -5
