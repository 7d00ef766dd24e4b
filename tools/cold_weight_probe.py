"""Is the decoder's M=960 GEMM slower in the step than in the A/B probe because its weights are cold?  A dependent chain of
960x768x768 GEMMs captured in a graph, (a) re-using ONE weight matrix, (b) walking through NW distinct ones (NW * 1.2 MB;
above the 256 MB Infinity Cache nothing survives from the previous replay), (c) as (b) with the next weight touched by a
tiny streaming kernel on a second stream one GEMM ahead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import ops
BF = torch.bfloat16
M, N, K, L = 960, 768, 768, 48

def run(nw, prefetch):
    a = torch.randn(M, K, device='cuda').to(BF)
    ws = [(torch.randn(N, K, device='cuda') * 0.03).to(BF) for _ in range(nw)]
    outs = [torch.empty(M, N, dtype=BF, device='cuda') for _ in range(2)]
    sink = torch.zeros(1, device='cuda')
    side = torch.cuda.Stream()
    def fn():
        x = a
        cur = torch.cuda.current_stream()
        for i in range(L):
            if prefetch and i + 1 < L:
                ev = torch.cuda.Event(); ev.record(cur); side.wait_event(ev)
                with torch.cuda.stream(side):
                    sink.add_(ws[(i + 1) % nw].view(-1)[::64].float().sum())      # touches every 128-B line
            x = ops.gemm(x, ws[i % nw], out=outs[i & 1])
        if prefetch:
            cur.wait_stream(side)
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    for flush in (False, True):
        best = 1e9
        for _ in range(5):
            if flush:
                JUNK.add_(1.0)                       # 1.2 GB of traffic: evicts L2 and the 256 MB Infinity Cache
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / L)
        print(f'weights={nw:4d} ({nw * N * K * 2 / 1e6:6.0f} MB) prefetch={int(prefetch)} cache_flushed={int(flush)}: {best:6.2f} us per GEMM', flush=True)

JUNK = torch.zeros(150 << 20, device='cuda')
for nw, pf in ((1, False), (48, False)):
    run(nw, pf)
