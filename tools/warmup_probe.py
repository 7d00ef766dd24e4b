"""Step time against the number of steps already run in the process (is bench.py's default warm-up long enough for the clocks / caches to settle?).
    python tools/warmup_probe.py"""
import sys
import time

sys.path.insert(0, '.')
import torch

import bench

tr, dims, _ = bench.build_trainer(32, True, 0)
torch.cuda.synchronize()
marks = [1, 5, 10, 20, 40, 80, 160, 320]
done, t_prev = 0, None
for m in marks:
    n = m - done
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step()
    torch.cuda.synchronize()
    print(f'steps {done + 1:4d}..{m:4d}: {(time.perf_counter() - t0) / n * 1e3:8.3f} ms/step', flush=True)
    done = m
time.sleep(3.0)                                       # an idle gap like the one between bench legs
for m in (5, 20, 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(m):
        tr.step()
    torch.cuda.synchronize()
    print(f'after 3 s idle, next {m:3d} steps: {(time.perf_counter() - t0) / m * 1e3:8.3f} ms/step', flush=True)
# host time of the enqueue alone (prologue + graph launch), against the device time it overlaps
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step()
h = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'host enqueue time {h / 20 * 1e3:8.3f} ms/step, with the device drained {(time.perf_counter() - t0) / 20 * 1e3:8.3f} ms/step', flush=True)
for k in (20, 20, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        tr.step()
    torch.cuda.synchronize()
    print(f'{k} steps: {(time.perf_counter() - t0) / k * 1e3:8.3f} ms/step', flush=True)
