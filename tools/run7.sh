#!/bin/bash
out=gpurun_out/r2_run7; mkdir -p $out
timeout 600 python tests/tools/diag_dropout.py > $out/diag_dropout.txt 2>&1; tail -12 $out/diag_dropout.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_inpaint.py -m gpu -q --timeout=600 -k "implicit or inpaint or saved or capped" > $out/pytest.log 2>&1; tail -15 $out/pytest.log
