"""Correctness + timing of the big-tile (256x128, LDS-DMA) GEMM kernel variants against the 128x128 register-staged kernel on
the forward shapes of the Prismer-BASE step.  One process, modes interleaved under hipGraph replay (ph_gemm_tuning switches).
    python tools/big_probe.py            -> table on stdout"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import _lib, ops
from prismer_amd._lib import ACT_QUICKGELU, ACT_RELU2

BF = torch.bfloat16
MODES = [int(m) for m in os.environ.get('BIG_MODES', '0,1,5').split(',')]
REPLAYS = int(os.environ.get('BIG_REPLAYS', '3'))      # graph replays per timing sample (10 launches each): raise for sustained-load numbers


def graph_of(fn, n):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    return g


def case(name, M, N, K, bias=True, act=0, pre=False, residual=False, f32res=False, out_f32=False, n=10, rounds=4, tb=False, act_in=False):
    a = (torch.randn(M, K, device='cuda') * 0.5).to(BF)
    b = (torch.randn(N, K, device='cuda') * 0.05).to(BF)
    kw = {}
    bt = b.t().contiguous() if tb else None          # [K, N]: the dgrad layout (dy . W with W = [N_out, K_in])
    if tb:
        kw['trans_b'] = True
    g_in = None
    if act_in:
        g_in = torch.randn(M, N, device='cuda').to(BF)
        kw['act_in'] = g_in
        kw['act'] = _lib.ACT_SAVED_GRAD
    if bias:
        kw['bias'] = torch.randn(N, device='cuda') * 0.1
    if act:
        kw['act'] = act
    if residual:
        kw['residual'] = torch.randn(M, N, device='cuda').to(torch.float32 if f32res else BF)
    out = torch.empty(M, N, dtype=torch.float32 if out_f32 else BF, device='cuda')
    off = int(os.environ.get('BIG_PRE_OFFSET', '0'))      # elements: shifts the second output's base relative to the first (channel-aliasing test)
    pre_t = torch.empty(M * N + off, dtype=BF, device='cuda')[off:].view(M, N) if pre else None
    if pre:
        kw['pre_out'] = pre_t
    # reference on a row sample (full fp32 matmul of 8320x3072x768 is fine on the GPU)
    z = a.float() @ b.float().t()
    if bias:
        z = z + kw['bias']
    zp = z.clone()
    if act == ACT_QUICKGELU:
        z = z * torch.sigmoid(1.702 * z)
    elif act == ACT_RELU2:
        z = torch.relu(z) ** 2
    if act_in:
        z = z * g_in.float()
    if residual:
        z = z + kw['residual'].float()
    graphs, errs = [], []
    bb = bt if tb else b
    for m in MODES:
        _lib.lib.ph_gemm_tuning(m, int(os.environ.get('BIG_MIN_TILES', '1')))      # 1: force the 256x128 kernel wherever eligible; 128: the step's dispatch
        out.zero_()
        ops.gemm(a, bb, out=out, out_f32=out_f32, **kw)
        torch.cuda.synchronize()
        e = ((out.float() - z).norm() / z.norm()).item()
        if pre:
            e = max(e, ((pre_t.float() - zp).norm() / zp.norm()).item())
        errs.append(e)
        graphs.append(graph_of(lambda: ops.gemm(a, bb, out=out, out_f32=out_f32, **kw), n))
    best = [1e30] * len(MODES)
    for _ in range(rounds):
        for i, g in enumerate(graphs):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(REPLAYS):
                g.replay()
            t1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], t0.elapsed_time(t1) / (REPLAYS * n) * 1e3)
            errs[i] = max(errs[i], ((out.float() - z).norm() / z.norm()).item())      # race screen: every replay batch is re-checked
    fl = 2.0 * M * N * K
    print(f'{name:30s} ' + ' | '.join(f'm{m}: {best[i]:6.1f}us {fl / best[i] / 1e6:5.0f}TF e={errs[i]:.1e}' for i, m in enumerate(MODES)), flush=True)
    assert all(e < 8e-3 for e in errs), errs


if __name__ == '__main__':
    torch.manual_seed(0)
    if os.environ.get('BIG_WIDE_ONLY'):          # the wide-N, short-K launches of the step
        case('vit fc 8320x3072x768 qgelu+pre', 8320, 3072, 768, act=ACT_QUICKGELU, pre=True)
        case('vit qkv 8320x2304x768', 8320, 2304, 768)
        case('dgrad proj 8320x3072x768 *g', 8320, 3072, 768, bias=False, tb=True, act_in=True)
        case('resampler kv 39680x1536x768', 39680, 1536, 768)
        case('cross kv all 8320x18432x768', 8320, 18432, 768)
        case('vit out 8320x768x768 +res', 8320, 768, 768, residual=True)
        _lib.lib.ph_gemm_tuning(*_lib.GEMM_BIG_DEFAULT)
        sys.exit(0)
    if os.environ.get('BIG_ONLY_FC'):
        case('vit fc 8320x3072x768 qgelu+pre', 8320, 3072, 768, act=ACT_QUICKGELU, pre=True)
        case('vit fc 8320x3072x768 qgelu (one output)', 8320, 3072, 768, act=ACT_QUICKGELU)
        sys.exit(0)
    case('plain 8192x4096x4096', 8192, 4096, 4096, bias=False)
    case('vit qkv 8320x2304x768', 8320, 2304, 768)
    case('vit out 8320x768x768 +res', 8320, 768, 768, residual=True)
    case('vit fc 8320x3072x768 qgelu+pre', 8320, 3072, 768, act=ACT_QUICKGELU, pre=True)
    case('vit proj 8320x768x3072 +res', 8320, 768, 3072, residual=True)
    case('resampler kv 39680x1536x768', 39680, 1536, 768)
    case('cross kv all 8320x18432x768', 8320, 18432, 768)
    case('stem 25088x384x1728', 25088, 384, 1728, bias=False)
    case('stem 6272x768x3456', 6272, 768, 3456, bias=False)
    case('dgrad proj 8320x3072x768 *g', 8320, 3072, 768, bias=False, tb=True, act_in=True)
    case('dgrad 8320x768x768 tb', 8320, 768, 768, bias=False, tb=True)
    case('dgrad fc 8320x768x3072 tb +res', 8320, 768, 3072, bias=False, tb=True, residual=True)
    case('dgrad qkv 8320x768x2304 tb', 8320, 768, 2304, bias=False, tb=True)
    case('dgrad kv 39680x768x1536 tb', 39680, 768, 1536, bias=False, tb=True)
    case('ragged 8300x2312x704', 8300, 2312, 704, act=ACT_RELU2, pre=True, residual=True)
    case('ragged tb 8300x2312x704', 8300, 2312, 704, act=ACT_RELU2, pre=True, residual=True, tb=True)
    case('f32 out/res 4160x1536x768', 4160, 1536, 768, residual=True, f32res=True, out_f32=True)
    _lib.lib.ph_gemm_tuning(*_lib.GEMM_BIG_DEFAULT)
