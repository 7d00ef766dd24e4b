"""small-GEMM probe under hipGraph replay (no host launch overhead): decoder shapes, forced split-K variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import ops
BF = torch.bfloat16

def graph_time(fn, n=40):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3

def run(M, N, K, layout, sks, **kw):
    if layout == 'nt':
        a = torch.randn(M, K, device='cuda').to(BF); b = torch.randn(N, K, device='cuda').to(BF); lk = {}
    elif layout == 'nn':
        a = torch.randn(M, K, device='cuda').to(BF); b = torch.randn(K, N, device='cuda').to(BF); lk = dict(trans_b=True)
    else:
        a = torch.randn(K, M, device='cuda').to(BF); b = torch.randn(K, N, device='cuda').to(BF); lk = dict(trans_a=True, trans_b=True)
    f32 = kw.get('out_f32', False)
    out = torch.zeros(M, N, dtype=torch.float32 if f32 else BF, device='cuda')
    res = []
    for sk in sks:
        t = graph_time(lambda: ops.gemm(a, b, out=out, split_k=sk, **lk, **kw))
        res.append(f'sk={sk}: {t:6.1f} us ({2.0*M*N*K/t/1e6:5.0f} TF)')
    print(f'{layout} M={M} N={N} K={K} {kw}: ' + ' | '.join(res), flush=True)

if __name__ == '__main__':
    run(960, 768, 768, 'nt', [0, 1, 2, 3])
    run(960, 2304, 768, 'nt', [0, 1, 2])
    run(960, 3072, 768, 'nt', [0, 1, 2])
    run(960, 768, 3072, 'nt', [0, 1, 2, 4, 8])
    run(960, 768, 768, 'nn', [0, 1, 2])
    run(768, 768, 960, 'tn', [0, 1, 2, 3, 5], out_f32=True, accumulate=True)
    run(3072, 768, 960, 'tn', [0, 1, 2, 3], out_f32=True, accumulate=True)
    run(8320, 768, 768, 'nt', [0, 1])
    run(8320, 3072, 768, 'nt', [0, 1])
