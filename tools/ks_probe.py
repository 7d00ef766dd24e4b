"""Decoder-shaped GEMMs (M = 960): correctness vs an fp32 matmul and per-launch time in a dependent chain under graph replay.
    python tools/ks_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import _lib, ops
from prismer_amd._lib import ACT_GELU

BF = torch.bfloat16


def chain(call, n=40):
    call(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            call()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def case(name, M, N, K, tb=False, bias=True, res32=False, out_f32=False, act=0, pre=False, act_in=False):
    a = (torch.randn(M, K, device='cuda') * 0.5).to(BF)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(BF)
    b = w.t().contiguous() if tb else w
    kw = dict(trans_b=True) if tb else {}
    z = a.float() @ w.float().t()
    if bias:
        kw['bias'] = torch.randn(N, device='cuda') * 0.1
        z = z + kw['bias']
    zp = None
    if act == ACT_GELU:
        kw['act'] = ACT_GELU
        if pre:
            kw['pre_out'] = torch.empty(M, N, dtype=BF, device='cuda'); kw['pre_grad'] = True
            x = z.double()
            zp = (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-x * x / 2) / (2 * 3.141592653589793) ** 0.5).float()
        z = torch.nn.functional.gelu(z)
    if act_in:
        gi = torch.randn(M, N, device='cuda').to(BF)
        kw['act_in'] = gi; kw['act'] = _lib.ACT_SAVED_GRAD
        z = z * gi.float()
    if res32:
        kw['residual'] = torch.randn(M, N, device='cuda')
        z = z + kw['residual']
    out = torch.empty(M, N, dtype=torch.float32 if out_f32 else BF, device='cuda')
    call = lambda: ops.gemm(a, b, out=out, out_f32=out_f32, **kw)
    call(); torch.cuda.synchronize()
    e = ((out.float() - z).norm() / z.norm()).item()
    if zp is not None:
        e = max(e, ((kw['pre_out'].float() - zp).norm() / zp.norm()).item())
    t = chain(call)
    e2 = ((out.float() - z).norm() / z.norm()).item()
    print(f'{name:44s} {t:7.2f} us  err {e:.1e} / {e2:.1e}', flush=True)
    assert max(e, e2) < 8e-3


if __name__ == '__main__':
    torch.manual_seed(0)
    case('dense 960x768x768 +bias +res32 -> f32', 960, 768, 768, res32=True, out_f32=True)
    case('q 960x768x768 +bias', 960, 768, 768)
    case('dgrad 960x768x768 tb', 960, 768, 768, tb=True, bias=False)
    case('mlp out 960x768x3072 +bias +res32 -> f32', 960, 768, 3072, res32=True, out_f32=True)
    case('dgrad qkv 960x768x2304 tb', 960, 768, 2304, tb=True, bias=False)
    case('dgrad mlp 960x768x3072 tb', 960, 768, 3072, tb=True, bias=False)
    case('dgrad saved 960x768x768 tb *g', 960, 768, 768, tb=True, bias=False, act_in=True)
    case('ragged 950x760x832', 950, 760, 832)
    case('ragged tb 950x760x832', 950, 760, 832, tb=True, bias=False)
    case('fc 960x3072x768 gelu+grad (64x64 kernel)', 960, 3072, 768, act=ACT_GELU, pre=True)
