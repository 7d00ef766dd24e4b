#!/bin/bash
out=gpurun_out/r2_run14; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_parity_gpu.py -m gpu -q --timeout=900 -k "dp_gpu or sharded or two_ranks or native_comm or pretrain_recipe or equals_eager" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $out/pytest.log | tail -20
