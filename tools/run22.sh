#!/bin/bash
out=gpurun_out/r2_run22; mkdir -p $out
timeout 420 python tools/big_probe.py > $out/big_probe.txt 2>&1; cat $out/big_probe.txt | tail -14
