timeout 900 python -m pytest tests/test_parity_gpu.py -x -q --timeout=600 -k "bench_gradient_handling" 2>&1 | tail -5
