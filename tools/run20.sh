#!/bin/bash
out=gpurun_out/r2_run20; mkdir -p $out
timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q --timeout=600 > $out/pytest.log 2>&1; tail -4 $out/pytest.log
PRISMER_DIST_BACKEND=gloo PRISMER_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $out/dryrun.json 2> $out/dryrun.err; tail -c 300 $out/dryrun.json
