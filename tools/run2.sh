#!/bin/bash
out=gpurun_out/r2_run2; mkdir -p $out
timeout 600 python tools/big_probe.py > $out/big_probe.txt 2>&1; cat $out/big_probe.txt | tail -20
