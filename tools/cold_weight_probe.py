"""Decoder-shaped GEMM chain (M = 960 rows) with HOT weights (one matrix re-used by every launch) against COLD weights (every launch reads a matrix
that left the caches: a ring of matrices larger than the 256-MB Infinity Cache) -- what does a launch of the step pay for its first touch of W?
    python tools/cold_weight_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import ops

BF = torch.bfloat16


def chain(M, N, K, nw, launches=64, reps=5):
    a = (torch.randn(M, K, device='cuda') * 0.5).to(BF)
    ws = [(torch.randn(N, K, device='cuda') * 0.05).to(BF) for _ in range(nw)]
    res = torch.randn(M, N, device='cuda').to(BF)
    out = torch.empty(M, N, dtype=BF, device='cuda')

    def run():
        for i in range(launches):
            ops.gemm(a, ws[i % nw], out=out, residual=res)
    run(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); g.replay(); t1.record(); torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / launches * 1e3)
    return best


if __name__ == '__main__':
    torch.manual_seed(0)
    for M, N, K in [(960, 768, 768), (960, 2304, 768), (960, 3072, 768), (960, 768, 3072), (8320, 768, 768), (8320, 3072, 768)]:
        wbytes = N * K * 2
        ring = max(2, int(600e6 // wbytes))
        hot, cold = chain(M, N, K, 1), chain(M, N, K, min(ring, 600))
        print(f'{M}x{N}x{K}: W = {wbytes / 1e6:.2f} MB   hot {hot:6.2f} us   cold (ring of {min(ring, 600)}) {cold:6.2f} us   first touch +{cold - hot:5.2f} us', flush=True)
