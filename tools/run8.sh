#!/bin/bash
out=gpurun_out/r2_run8; mkdir -p $out
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=600 -k "equals_eager or autograd_path" > $out/iso.log 2>&1; tail -4 $out/iso.log
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=600 -k "zbase_b4 and hipgraph or equals_eager or autograd_path" > $out/after_graph.log 2>&1; tail -4 $out/after_graph.log
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=600 -k "micro_batch or equals_eager or autograd_path" > $out/after_micro.log 2>&1; tail -4 $out/after_micro.log
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=600 -k "test_trainer_step_matches_oracle_adamw or equals_eager or autograd_path" > $out/after_adamw.log 2>&1; tail -4 $out/after_adamw.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "implicit" > $out/implicit.log 2>&1; tail -8 $out/implicit.log
