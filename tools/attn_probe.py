"""Attention kernels at the step's shapes: correctness against an fp32 torch reference + timing under hipGraph replay.
    ATTN_TUNING=0|1|2 python tools/attn_probe.py      (ph_attention_tuning: 0 streaming kernels only, 1 default, 2 default without the one-pass dQ kernel)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import ops, _lib

BF = torch.bfloat16


def graph_time(fn, n=10, reps=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); g.replay(); t1.record(); torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / n * 1e3)
    return best


def case(name, B, H, Sq, Sk, dh):
    W = H * dh
    q = (torch.randn(B * Sq, W, device='cuda') * 1.0).to(BF)
    kv = (torch.randn(B * Sk, 2 * W, device='cuda') * 1.0).to(BF)
    do = torch.randn(B * Sq, W, device='cuda').to(BF)
    ks = (Sk * 2 * W, 2 * W)
    qs = (Sq * W, W)
    o, lse = ops.attention_fwd(q, kv[:, :W], kv[:, W:], B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks)
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)

    def bwd():
        ops.attention_bwd(do, q, kv[:, :W], kv[:, W:], o, lse, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, dq=dq,
                          dk=dkv[:, :W], dv=dkv[:, W:], dq_strides=qs, dk_strides=ks, dv_strides=ks)
    bwd(); torch.cuda.synchronize()
    # fp32 reference
    qf = q.float().view(B, Sq, H, dh).transpose(1, 2).requires_grad_(True)
    kf = kv[:, :W].float().reshape(B, Sk, H, dh).transpose(1, 2).requires_grad_(True)
    vf = kv[:, W:].float().reshape(B, Sk, H, dh).transpose(1, 2).requires_grad_(True)
    sc = (qf @ kf.transpose(-1, -2)) * dh ** -0.5
    p = torch.softmax(sc, -1)
    of = p @ vf
    of.backward(do.float().view(B, Sq, H, dh).transpose(1, 2))
    rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
    e_o = rel(o.view(B, Sq, H, dh).transpose(1, 2), of.detach())
    e_l = rel(lse.view(B, H, Sq), torch.logsumexp(sc.detach(), -1))
    e_dq = rel(dq.view(B, Sq, H, dh).transpose(1, 2), qf.grad)
    e_dk = rel(dkv[:, :W].reshape(B, Sk, H, dh).transpose(1, 2), kf.grad)
    e_dv = rel(dkv[:, W:].reshape(B, Sk, H, dh).transpose(1, 2), vf.grad)
    t_f = graph_time(lambda: ops.attention_fwd(q, kv[:, :W], kv[:, W:], B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, out=o))
    t_b = graph_time(bwd)
    fl = 4.0 * B * H * Sq * Sk * dh
    print(f'{name:28s} fwd {t_f:7.1f} us {fl / t_f / 1e6:5.0f} TF | bwd {t_b:7.1f} us {2.5 * fl / t_b / 1e6:5.0f} TF | err o {e_o:.1e} lse {e_l:.1e} dq {e_dq:.1e} dk {e_dk:.1e} dv {e_dv:.1e}',
          flush=True)
    assert max(e_o, e_dq, e_dk, e_dv) < 2e-2 and e_l < 1e-4


if __name__ == '__main__':
    torch.manual_seed(0)
    _lib.lib.ph_attention_tuning(int(os.environ.get('ATTN_TUNING', '1')))
    print('ph_attention_tuning =', os.environ.get('ATTN_TUNING', '1'))
    case('vit 32x12 S=260 dh=64', 32, 12, 260, 260, 64)
    case('resampler 32x8 64x1240 dh=96', 32, 8, 64, 1240, 96)
    case('zbase 32x12 S=196 dh=64', 32, 12, 196, 196, 64)
    case('large 4x16 S=1220 dh=64', 4, 16, 1220, 1220, 64)
    case('cross 32x12 30x260 dh=64', 32, 12, 30, 260, 64)
