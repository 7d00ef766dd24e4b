#!/bin/bash
out=gpurun_out/r2_run17; mkdir -p $out
AB_LIBS=prismer_amd/lib/libprismer_hip.so,prismer_amd/lib/libprismer_hip_adamwnt.so timeout 300 python tools/adamw_probe.py > $out/adamw.txt 2>&1; tail -7 $out/adamw.txt
