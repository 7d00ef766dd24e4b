"""Micro-benchmarks of the hot kernels at Prismer-BASE bs32 shapes (run on the GPU box).
python tools/kernel_bench.py [out.json]"""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from prismer_amd import ops

BF = torch.bfloat16


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    res = []
    shapes = [  # name, M, N, K, layout
        ('vit qkv', 8320, 2304, 768, 'nt'), ('vit out/adaptor', 8320, 768, 768, 'nt'), ('vit c_fc', 8320, 3072, 768, 'nt'),
        ('vit c_proj', 8320, 768, 3072, 'nt'), ('resampler kv', 39680, 1536, 768, 'nt'), ('dec qkv', 960, 2304, 768, 'nt'),
        ('dec dense', 960, 768, 768, 'nt'), ('dec mlp up', 960, 3072, 768, 'nt'), ('lm head', 960, 50304, 768, 'nt'),
        ('dgrad c_fc', 8320, 768, 3072, 'nn'), ('dgrad qkv', 8320, 768, 2304, 'nn'), ('dgrad kv', 39680, 768, 1536, 'nn'),
        ('wgrad adaptor', 768, 768, 8320, 'tn'), ('wgrad kv', 1536, 768, 39680, 'tn'), ('wgrad dec', 768, 768, 960, 'tn'),
        ('wgrad lm', 50304, 768, 960, 'tn'), ('stem conv2', 100352, 192, 864, 'nt'), ('stem conv1', 401408, 96, 32, 'nt'),
        ('square 4096', 4096, 4096, 4096, 'nt'),
    ]
    for name, M, N, K, lay in shapes:
        if lay == 'nt':
            a = torch.randn(M, K, device='cuda').to(BF); b = torch.randn(N, K, device='cuda').to(BF)
            out = torch.empty(M, N, dtype=BF, device='cuda')
            fn = lambda: ops.gemm(a, b, out=out)
        elif lay == 'nn':
            a = torch.randn(M, K, device='cuda').to(BF); b = torch.randn(K, N, device='cuda').to(BF)
            out = torch.empty(M, N, dtype=BF, device='cuda')
            fn = lambda: ops.gemm(a, b, out=out, trans_b=True)
        else:
            a = torch.randn(K, M, device='cuda').to(BF); b = torch.randn(K, N, device='cuda').to(BF)
            out = torch.zeros(M, N, dtype=torch.float32, device='cuda')
            fn = lambda: ops.gemm(a, b, out=out, trans_a=True, trans_b=True, out_f32=True, accumulate=True)
        us = timeit(fn)
        tf = 2.0 * M * N * K / us / 1e6
        res.append(dict(kernel='gemm', name=name, M=M, N=N, K=K, layout=lay, us=round(us, 2), tflops=round(tf, 1)))
        print(res[-1], flush=True)
    # layernorm
    x = torch.randn(8320, 768, device='cuda').to(BF); g = torch.ones(768, device='cuda'); b = torch.zeros(768, device='cuda')
    y = torch.empty_like(x)
    us = timeit(lambda: ops.layernorm_fwd(x, g, b, out=y))
    res.append(dict(kernel='layernorm_fwd', M=8320, D=768, us=round(us, 2), gbps=round(2 * x.numel() * 2 / us / 1e3, 1)))
    print(res[-1])
    # attention (ViT shape)
    B, H, S, dh = 32, 12, 260, 64
    qkv = torch.randn(B * S, 3 * H * dh, device='cuda').to(BF)
    D = H * dh
    st = (S * 3 * D, 3 * D)
    us = timeit(lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, S, S, dh, q_strides=st, k_strides=st, v_strides=st))
    res.append(dict(kernel='attn_fwd vit', us=round(us, 2), tflops=round(4.0 * B * H * S * S * dh / us / 1e6, 1)))
    print(res[-1])
    B, H, Sq, Sk, dh = 32, 8, 64, 1240, 96
    D = H * dh
    q = torch.randn(B * Sq, D, device='cuda').to(BF); kv = torch.randn(B * Sk, 2 * D, device='cuda').to(BF)
    us = timeit(lambda: ops.attention_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, dh, q_strides=(Sq * D, D), k_strides=(Sk * 2 * D, 2 * D), v_strides=(Sk * 2 * D, 2 * D)))
    res.append(dict(kernel='attn_fwd perceiver', us=round(us, 2), tflops=round(4.0 * B * H * Sq * Sk * dh / us / 1e6, 1)))
    print(res[-1])
    # adamw
    n = 64 * 1024 * 1024
    p = torch.randn(n, device='cuda'); gr = torch.randn(n, device='cuda'); m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
    pb = torch.empty(n, dtype=BF, device='cuda'); hy = torch.tensor([5e-5, 0.1, 0.001], device='cuda')
    us = timeit(lambda: ops.adamw(p, gr, m, v, pb, n, hy))
    res.append(dict(kernel='adamw', n=n, us=round(us, 2), gbps=round(n * 30 / us / 1e3, 1)))
    print(res[-1])
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
