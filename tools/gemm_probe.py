"""GEMM fixed-cost / per-k-tile slope probe (MI355X): time vs K for forced tile sizes, hot (same buffers) and cold (rotating buffers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import ops
BF = torch.bfloat16

def timeit(fns, iters=30, warm=5):
    for i in range(warm): fns[i % len(fns)]()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

def probe(M, N, layout, Ks, force):
    for K in Ks:
        sets = []
        for r in range(6):          # rotate 6 buffer sets: > 256 MB total for big shapes -> cold-ish
            if layout == 'nt':
                a = torch.randn(M, K, device='cuda').to(BF); b = torch.randn(N, K, device='cuda').to(BF)
                kw = {}
            elif layout == 'nn':
                a = torch.randn(M, K, device='cuda').to(BF); b = torch.randn(K, N, device='cuda').to(BF); kw = dict(trans_b=True)
            else:
                a = torch.randn(K, M, device='cuda').to(BF); b = torch.randn(K, N, device='cuda').to(BF); kw = dict(trans_a=True, trans_b=True)
            out = torch.empty(M, N, dtype=BF, device='cuda')
            sets.append((a, b, out, kw))
        res = []
        for sk in force:
            hot = timeit([lambda s=sets[0]: ops.gemm(s[0], s[1], out=s[2], split_k=sk, **s[3])])
            cold = timeit([(lambda s=s: ops.gemm(s[0], s[1], out=s[2], split_k=sk, **s[3])) for s in sets])
            res.append(f'sk={sk}: hot {hot:7.1f} us cold {cold:7.1f} us ({2.0*M*N*K/cold/1e6:6.0f} TF)')
        print(f'{layout} M={M} N={N} K={K}: ' + ' | '.join(res), flush=True)

if __name__ == '__main__':
    # split_k=0 -> cost model ; split_k=1 -> forces 128x128 (when >= 128 tiles) without split
    probe(8320, 768, 'nt', [64, 768, 3072], [0, 1])
    probe(8320, 3072, 'nt', [64, 768], [0, 1])
    probe(8320, 768, 'nn', [768, 3072], [0, 1])
    probe(960, 768, 'nt', [64, 768, 3072], [0, 1])
    probe(768, 768, 'tn', [960, 8320], [0, 1, 8])
