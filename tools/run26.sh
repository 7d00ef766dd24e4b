#!/bin/bash
out=gpurun_out/r2_run26; mkdir -p $out
BIG_MODES=0,5 BIG_REPLAYS=300 timeout 500 python tools/big_probe.py > $out/big_probe_sustained.txt 2>&1; cat $out/big_probe_sustained.txt | tail -18
rocm-smi --showclocks --showpower 2>/dev/null | head -30
