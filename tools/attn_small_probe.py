"""A/B of the small-query attention kernels against the streaming kernels on the decoder's launches (Prismer-BASE, batch 32, T = 30, S = 260):
HIP-event time per launch over graph-free back-to-back launches, both families, forward and backward.  `python tools/attn_small_probe.py`"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prismer_amd import _lib, ops


def bench(fn, n=50, reps=20):
    """device time per launch: n launches captured into one hipGraph (a dependent chain on one stream, like the decoder's), replayed"""
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
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


def main():
    B, H, dh, T = 32, 12, 64, 30
    D = H * dh
    seed = torch.tensor([1234], dtype=torch.int64, device='cuda')
    for name, Sk, causal, masked, B_, T_ in (('self  T=30', 30, True, True, B, T), ('cross S=260', 260, False, False, B, T), ('self  T=40 (VQA, B=16, H=16)', 40, True, True, 16, 40)):
        Hh = 16 if 'VQA' in name else H
        Dd = Hh * dh
        q = torch.randn(B_ * T_, Dd, device='cuda').bfloat16()
        kv = torch.randn(B_ * Sk, 2 * Dd, device='cuda').bfloat16()
        k, v = kv[:, :Dd], kv[:, Dd:]
        km = torch.ones(B_, Sk, dtype=torch.uint8, device='cuda') if masked else None
        drop = ops.Dropout(0.1, seed, 7)
        qs, ks = (T_ * Dd, Dd), (Sk * 2 * Dd, 2 * Dd)
        d_o = torch.randn(B_ * T_, Dd, device='cuda').bfloat16()
        dq = torch.empty_like(q); dkv = torch.empty_like(kv)
        res = {}
        for fam in (1, 0):
            _lib.lib.ph_attention_tuning(fam)
            o, lse = ops.attention_fwd(q, k, v, B_, Hh, T_, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, key_mask=km, causal=causal, drop=drop)
            f = bench(lambda: ops.attention_fwd(q, k, v, B_, Hh, T_, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, key_mask=km, causal=causal, drop=drop))
            bw = bench(lambda: ops.attention_bwd(d_o, q, k, v, o, lse, B_, Hh, T_, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, dq=dq, dk=dkv[:, :Dd],
                                                 dv=dkv[:, Dd:], dq_strides=qs, dk_strides=ks, dv_strides=ks, key_mask=km, causal=causal, drop=drop))
            res[fam] = (f, bw)
        _lib.lib.ph_attention_tuning(1)
        print(f'{name:32s} forward {res[0][0]:6.1f} -> {res[1][0]:6.1f} us   backward (dQ + dK/dV) {res[0][1]:6.1f} -> {res[1][1]:6.1f} us   (streaming -> small-query kernels, per launch inside a replayed hipGraph chain)')


if __name__ == '__main__':
    main()
