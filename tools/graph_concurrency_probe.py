"""Does this ROCm's hipGraph executor overlap independent branches?  Two chains of small GEMMs (each fills ~18 % of the chip)
captured (a) back to back on one stream, (b) as two parallel branches; plus the same with plain stream launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import ops
BF = torch.bfloat16
M, N, K, L = 960, 768, 768, 24

def mk():
    a = torch.randn(M, K, device='cuda').to(BF); w = torch.randn(N, K, device='cuda').to(BF) * 0.03
    return a, w, [torch.empty(M, N, dtype=BF, device='cuda') for _ in range(2)]
A, B = mk(), mk()

def chain(c):
    a, w, o = c
    x = a
    for i in range(L):
        x = ops.gemm(x, w, out=o[i & 1])

def serial():
    chain(A); chain(B)

S1, S2 = torch.cuda.Stream(), torch.cuda.Stream()
def parallel():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    S1.wait_event(ev); S2.wait_event(ev)
    with torch.cuda.stream(S1): chain(A)
    with torch.cuda.stream(S2): chain(B)
    cur.wait_stream(S1); cur.wait_stream(S2)

def timed(fn, graph):
    fn(); torch.cuda.synchronize()
    if graph:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): fn()
        run = g.replay
    else:
        run = fn
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3)
    return best

for name, fn in (('serial', serial), ('parallel', parallel)):
    for graph in (True, False):
        t = timed(fn, graph)
        print(f'{name:9s} graph={int(graph)}: {t:8.1f} us total, {t / (2 * L):6.2f} us per GEMM', flush=True)
