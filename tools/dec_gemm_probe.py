"""Device time per launch of the decoder's GEMMs inside a replayed hipGraph CHAIN of dependent launches (how the decoder runs them), on its
shapes and fused chains (profiles/r5_decoder_gemm.txt; the two kernel forms A/B-ed there in round 5 are in the history, not in the product).
`python tools/dec_gemm_probe.py`"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prismer_amd import _lib, ops
from prismer_amd._lib import ACT_RELU2

BF = torch.bfloat16


def chain_time(fn, n=40, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


def main():
    M = 960
    seed = torch.tensor([77], dtype=torch.int64, device='cuda')
    rows = []
    for name, N, K, tb, kind in (('q / cross-q projection      960x768x768   bias', 768, 768, False, 'plain'),
                                 ('out-projection              960x768x768   bias+dropout+fp32 residual -> fp32', 768, 768, False, 'post'),
                                 ('MLP output                  960x768x3072  bias+dropout+fp32 residual -> fp32', 768, 3072, False, 'post'),
                                 ('adaptor down                960x384x768   bias+relu^2 + derivative', 384, 768, False, 'relu2'),
                                 ('data gradient [K,N]         960x768x768   + bf16 skip gradient', 768, 768, True, 'res'),
                                 ('data gradient [K,N]         960x768x3072  plain', 768, 3072, True, 'none')):
        a = torch.randn(M, K, device='cuda').to(BF)
        b = (torch.randn(K, N, device='cuda') if tb else torch.randn(N, K, device='cuda')).to(BF) * 0.05
        bias = torch.randn(N, device='cuda')
        resf, resb = torch.randn(M, N, device='cuda'), torch.randn(M, N, device='cuda').to(BF)
        outf, outb, pre = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda', dtype=BF), torch.empty(M, N, device='cuda', dtype=BF)
        d = ops.Dropout(0.1, seed, 5)
        fn = {'plain': lambda: ops.gemm(a, b, out=outb, trans_b=tb, bias=bias),
              'post': lambda: ops.gemm(a, b, out=outf, trans_b=tb, bias=bias, drop=d, residual=resf, out_f32=True),
              'relu2': lambda: ops.gemm(a, b, out=outb, trans_b=tb, bias=bias, act=ACT_RELU2, pre_out=pre, pre_grad=True),
              'res': lambda: ops.gemm(a, b, out=outb, trans_b=tb, residual=resb),
              'none': lambda: ops.gemm(a, b, out=outb, trans_b=tb)}[kind]
        print(f'{name:92s} {chain_time(fn):6.2f} us per launch')


if __name__ == '__main__':
    main()
