"""GPU: every HIP kernel (through the C ABI) against a plain PyTorch fp32 reference of the same op.
Tolerances: bf16 outputs carry one rounding (2^-9 relative) on top of fp32-accumulated math: rel-Frobenius
<= 6e-3; fp32 outputs of bf16-input contractions <= 2e-4."""
import math

import numpy as np
import os
import pytest
import torch
import torch.nn.functional as F

from tests.util import bf16_round, dropout_keep_attention, dropout_keep_linear, max_abs, rel_fro

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope='module')
def ops():
    from prismer_amd import ops as o
    return o


def rnd(*shape, scale=1.0, seed=0, dtype=BF):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def seed_tensor(val):
    return torch.tensor([val], dtype=torch.int64, device='cuda')


# ---------------------------------------------------------------------------------------------- layouts
def test_probe_layouts(ops):
    inp = (torch.arange(1024) % 256).to(BF).cuda()
    out = ops.probe_layouts(inp).cpu()
    src = (torch.arange(1024) % 256).float()
    tr = out[:256].reshape(64, 4)
    exp = torch.empty(64, 4)
    for l in range(64):
        for j in range(4):
            exp[l, j] = src[(l & 15) + j * 16 + (l >> 4) * 64]       # guide: lane l, elem j of a lane-linear tr read
    assert torch.equal(tr, exp), f'ds_read_b64_tr_b16 semantics differ:\n got {tr[:20].tolist()}\n exp {exp[:20].tolist()}'
    c16 = out[256:512].reshape(64, 4)
    e16 = torch.empty(64, 4)
    for l in range(64):
        for r in range(4):
            e16[l, r] = src[((l >> 4) * 4 + r) * 16 + (l & 15)]
    assert torch.equal(c16, e16), 'mfma 16x16x32 C layout differs'
    c32 = out[512:].reshape(64, 16)
    e32 = torch.zeros(64, 16)
    for l in range(64):
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
            if row < 16:
                e32[l, r] = src[row * 32 + (l & 31)]
    assert torch.equal(c32, e32), 'mfma 32x32x16 C layout differs'


# ---------------------------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [(128, 128, 64), (200, 136, 96), (960, 768, 768), (1000, 2304, 768), (77, 1003, 256), (4160, 768, 3072),
               (300, 96, 32), (64, 64, 64)]


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
@pytest.mark.parametrize('layout', ['nt', 'nn', 'tn'])
def test_gemm_layouts(ops, M, N, K, layout):
    Np = (N + 7) // 8 * 8
    Mp = (M + 7) // 8 * 8
    a = rnd(M, K, scale=0.5, seed=1)
    b = rnd(N, K, scale=0.5, seed=2)
    ref = a.float() @ b.float().t()
    if layout == 'nt':
        if K % 8:
            pytest.skip('NT needs K % 8 == 0')
        out = ops.gemm(a, b)
    elif layout == 'nn':          # B stored [K][N] (padded leading dim)
        bt = torch.zeros(K, Np, dtype=BF, device='cuda'); bt[:, :N] = b.t()
        out = ops.gemm(a, bt, trans_b=True, N=N)
    else:                         # both stored [K][rows]
        at = torch.zeros(K, Mp, dtype=BF, device='cuda'); at[:, :M] = a.t()
        bt = torch.zeros(K, Np, dtype=BF, device='cuda'); bt[:, :N] = b.t()
        out = ops.gemm(at, bt, trans_a=True, trans_b=True, M=M, N=N)
    assert out.shape == (M, N)
    assert rel_fro(out, ref) < 6e-3, (layout, M, N, K, rel_fro(out, ref))


@pytest.mark.parametrize('act', [0, 1, 2, 3, 4])
def test_gemm_epilogue(ops, act):
    M, N, K = 333, 520, 256
    a, b = rnd(M, K, scale=0.5, seed=3), rnd(N, K, scale=0.3, seed=4)
    bias = rnd(N, dtype=torch.float32, seed=5)
    res = rnd(M, N, seed=6)
    pre = torch.empty(M, N, dtype=BF, device='cuda')
    out = ops.gemm(a, b, bias=bias, act=act, pre_out=pre, residual=res)
    z = a.float() @ b.float().t() + bias
    fn = {0: lambda t: t, 1: lambda t: t * torch.sigmoid(1.702 * t), 2: lambda t: torch.relu(t) ** 2,
          3: lambda t: F.gelu(t), 4: torch.relu}[act]
    assert rel_fro(pre, z) < 6e-3
    assert rel_fro(out, fn(z) + res.float()) < 6e-3
    # backward-through-activation epilogue: C = acc * act'(act_in)
    dy = rnd(M, N, scale=0.5, seed=7)
    w2 = rnd(N, K, scale=0.3, seed=8)          # dY[M,N] . W[N,K] -> [M,K], times act'(p) with p [M,K]
    p = rnd(M, K, seed=9)
    got = ops.gemm(dy, w2, trans_b=True, act=act, act_in=p)
    pf = p.float().requires_grad_(True)
    fn(pf).backward(torch.ones_like(pf))
    ref = (dy.float() @ w2.float()) * pf.grad
    assert rel_fro(got, ref) < 6e-3


@pytest.mark.parametrize('act', [1, 2, 3])
def test_gemm_saved_activation_gradient(ops, act):
    """pre_grad: the forward epilogue stores act'(x) (from the fp32 accumulator) instead of x; the backward GEMM with
    PH_ACT_SAVED_GRAD multiplies by it.  Together = the autograd gradient through Linear -> act."""
    from prismer_amd._lib import ACT_SAVED_GRAD
    M, N, K = 333, 520, 256
    a, b = rnd(M, K, scale=0.5, seed=3), rnd(N, K, scale=0.3, seed=4)
    bias = rnd(N, dtype=torch.float32, seed=5)
    g = torch.empty(M, N, dtype=BF, device='cuda')
    out = ops.gemm(a, b, bias=bias, act=act, pre_out=g, pre_grad=True)
    fn = {1: lambda t: t * torch.sigmoid(1.702 * t), 2: lambda t: torch.relu(t) ** 2, 3: lambda t: F.gelu(t)}[act]
    z = (a.float() @ b.float().t() + bias).requires_grad_(True)
    y = fn(z)
    y.backward(torch.ones_like(y))
    assert rel_fro(out, y) < 6e-3
    assert rel_fro(g, z.grad) < 6e-3
    dy = rnd(M, 64, scale=0.5, seed=7)
    w2 = rnd(64, N, scale=0.3, seed=8)           # dY[M,64] . W[64,N] -> [M,N], times the saved derivative
    got = ops.gemm(dy, w2, trans_b=True, act=ACT_SAVED_GRAD, act_in=g)
    assert rel_fro(got, (dy.float() @ w2.float()) * g.float()) < 6e-3


def test_gemm_grouped_capped_background_launch(ops):
    """ph_gemm_grouped_capped_bf16: a grid smaller than the tile count (blocks walk several tiles) gives the same result"""
    import ctypes as C
    from prismer_amd import _lib
    shapes = [(768, 768, 960), (2304, 768, 960), (104, 200, 960)]
    ops_, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, x = rnd(K, M, scale=0.3, seed=20 + i), rnd(K, N, scale=0.3, seed=40 + i)
        ops_.append((dy, x, torch.zeros(M, N, device='cuda')))
        refs.append(dy.float().t() @ x.float())
    for cap in (0, 7, 40, 100000):
        arr = (_lib.GemmArgs * len(shapes))()
        for g, (dy, x, gw) in zip(arr, ops_):
            gw.zero_()
            g.A, g.B, g.C = dy.data_ptr(), x.data_ptr(), gw.data_ptr()
            g.M, g.N, g.K = gw.shape[0], gw.shape[1], dy.shape[0]
            g.lda, g.ldb, g.ldc = dy.stride(0), x.stride(0), gw.stride(0)
            g.trans_a, g.trans_b, g.out_f32, g.accumulate, g.alpha = 1, 1, 1, 1, 1.0
        _lib.check(_lib.lib.ph_gemm_grouped_capped_bf16(arr, len(shapes), cap, torch.cuda.current_stream().cuda_stream), 'capped')
        for (dy, x, gw), ref in zip(ops_, refs):
            assert rel_fro(gw, ref) < 3e-4, cap


@pytest.mark.parametrize('trans_b', [False, True])
@pytest.mark.parametrize('mode', [1, 6, 7])
@pytest.mark.parametrize('M,N,K,act', [(1000, 776, 640, 1), (2048, 256, 128, 0), (515, 1536, 3072, 3)])
def test_gemm_big_tile_lds_dma_kernel(ops, M, N, K, act, mode, trans_b):
    """the 256x128 LDS-DMA kernel (global_load_lds into a 3-stage ring, counted vmcnt, raw barriers; mode 1 = plain main loop, 6 =
    ping-pong phases of the two waves of a SIMD with the LEAN tail, 7 = + the DMA requests spread over the M phase, the default), forced for shapes it would not normally take: ragged M / N
    tails, short and long k loops, the full fused epilogue, B given as [N,K] and as [K,N] -- against fp32 torch and against the
    128x128 register-staged kernel on the same inputs; repeated, because a mis-ordered DMA ring shows up as a rare wrong tile."""
    from prismer_amd import _lib
    a, b = rnd(M, K, scale=0.5, seed=3), rnd(N, K, scale=0.2, seed=4)
    bb = b.t().contiguous() if trans_b else b
    bias = rnd(N, dtype=torch.float32, seed=5)
    res = rnd(M, N, seed=6)
    pre_big = torch.empty(M, N, dtype=BF, device='cuda'); pre_old = torch.empty_like(pre_big)
    try:
        _lib.lib.ph_gemm_tuning(0, 128)
        old = ops.gemm(a, bb, trans_b=trans_b, bias=bias, act=act, pre_out=pre_old, residual=res)
        _lib.lib.ph_gemm_tuning(mode, 1)
        z = a.float() @ b.float().t() + bias
        fn = {0: lambda t: t, 1: lambda t: t * torch.sigmoid(1.702 * t), 3: lambda t: F.gelu(t)}[act]
        for rep in range(6):
            pre_big.zero_()
            big = ops.gemm(a, bb, trans_b=trans_b, bias=bias, act=act, pre_out=pre_big, residual=res)
            assert rel_fro(pre_big, z) < 6e-3 and rel_fro(big, fn(z) + res.float()) < 6e-3, rep
            assert rel_fro(big, old.float()) < 2e-3 and rel_fro(pre_big, pre_old.float()) < 2e-3, rep
    finally:
        _lib.lib.ph_gemm_tuning(*_lib.GEMM_BIG_DEFAULT)


@pytest.mark.parametrize('shapes', [
    [(1536, 768, 4096)] * 4,                         # 144 tiles of 256x128, 64 k-tiles: the resampler K/V weight-gradient pattern
    [(1544, 776, 2048)] * 4,                         # ragged M / N edges
    [(768, 384, 8320)] * 16,                         # 16 adaptor weight gradients
])
def test_gemm_grouped_big_tile_wgrad(shapes, ops):
    """weight-gradient groups with long reductions run on the 256x128 ping-pong kernel (A = [K,M] and B = [K,N] both through the
    LDS-DMA ring and the transposing fragment reads; persistent one-block-per-CU grid): dW += dY^T X against fp32 torch and against
    the 128x128 grouped kernel (capped entry point with a huge cap), on a non-zero accumulator, repeated."""
    from prismer_amd import _lib
    ops_, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, x = rnd(K, M, scale=0.3, seed=20 + i), rnd(K, N, scale=0.3, seed=40 + i)
        ops_.append((dy, x, torch.empty(M, N, device='cuda')))
        refs.append(dy.float().t() @ x.float())
    c0 = [torch.randn_like(o[2]) for o in ops_]
    arr = (_lib.GemmArgs * len(shapes))()
    for g, (dy, x, gw) in zip(arr, ops_):
        g.A, g.B, g.C = dy.data_ptr(), x.data_ptr(), gw.data_ptr()
        g.M, g.N, g.K = gw.shape[0], gw.shape[1], dy.shape[0]
        g.lda, g.ldb, g.ldc = dy.stride(0), x.stride(0), gw.stride(0)
        g.trans_a, g.trans_b, g.out_f32, g.accumulate, g.alpha = 1, 1, 1, 1, 1.0
    results = []
    for cap, reps in ((0, 4), (1 << 30, 1)):
        for rep in range(reps):
            for (dy, x, gw), c in zip(ops_, c0):
                gw.copy_(c)
            _lib.check(_lib.lib.ph_gemm_grouped_capped_bf16(arr, len(shapes), cap, torch.cuda.current_stream().cuda_stream), 'grouped')
            for (dy, x, gw), ref, c in zip(ops_, refs, c0):
                assert rel_fro(gw - c, ref) < 3e-4, (cap, rep, rel_fro(gw - c, ref))
        results.append([o[2].clone() for o in ops_])
    for x, y in zip(*results):
        assert rel_fro(x, y) < 1e-5
    # single-problem entry point, same layout
    dy, x, gw = ops_[0]
    gw.copy_(c0[0])
    ops.gemm(dy, x, out=gw, trans_a=True, trans_b=True, out_f32=True, accumulate=True)
    assert rel_fro(gw - c0[0], refs[0]) < 3e-4


@pytest.mark.parametrize('M,N,K,tb', [(960, 768, 768, False), (960, 768, 3072, True), (960, 2304, 832, False), (200, 136, 512, True)])
def test_gemm_intra_block_k_split(ops, M, N, K, tb):
    """gemm_ks2_kernel (64x64 tiles, 512 threads, two thread groups on alternate k-tiles, partial tiles summed by the write-out): the
    decoder-sized launches take it by default.  Against fp32 torch with the full fused epilogue, odd and even k-tile counts, and
    bit-for-bit repeatable (fixed summation order: no atomics)."""
    a, b = rnd(M, K, scale=0.5, seed=3), rnd(N, K, scale=0.2, seed=4)
    bb = b.t().contiguous() if tb else b
    bias, res = rnd(N, dtype=torch.float32, seed=5), rnd(M, N, dtype=torch.float32, seed=6)
    z = F.gelu(a.float() @ b.float().t() + bias) + res
    outs = [ops.gemm(a, bb, trans_b=tb, bias=bias, act=3, residual=res) for _ in range(3)]
    assert rel_fro(outs[0], z) < 6e-3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_gemm_f32_accumulate_splitk(ops):
    M, N, K = 768, 768, 4160           # wgrad shape: dW[N_out, K_in] = dY^T X, reduction over 4160 rows
    dy, x = rnd(K, M, scale=0.3, seed=10), rnd(K, N, scale=0.3, seed=11)
    ref = dy.float().t() @ x.float()
    c0 = torch.randn(M, N, device='cuda')
    for split in (1, 0, 7):
        c = c0.clone()
        ops.gemm(dy, x, out=c, trans_a=True, trans_b=True, out_f32=True, accumulate=True, split_k=split)
        assert rel_fro(c - c0, ref) < 3e-4, (split, rel_fro(c - c0, ref))
    c = ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True)
    assert rel_fro(c, ref) < 3e-4
    # bf16 accumulate
    cb = rnd(M, N, seed=12)
    cb0 = cb.clone()
    ops.gemm(dy, x, out=cb, trans_a=True, trans_b=True, accumulate=True, alpha=0.5)
    assert rel_fro(cb, cb0.float() + 0.5 * ref) < 6e-3
    # tiny output, very long reduction (the stems' first-layer weight gradients: 96 x 32 over 401 408 rows): up to 256 splits, folded by
    # the 16-lanes-per-vector reduce (round 3)
    M2, N2, K2 = 96, 32, 100352
    dy2, x2 = rnd(K2, M2, scale=0.3, seed=13), rnd(K2, N2, scale=0.3, seed=14)
    ref2 = dy2.float().t() @ x2.float()
    for split in (0, 64, 256):
        c2 = ops.gemm(dy2, x2, trans_a=True, trans_b=True, out_f32=True, split_k=split)
        assert rel_fro(c2, ref2) < 3e-4, (split, rel_fro(c2, ref2))


def test_wgrad_groups_split_over_k_match_single_launches(ops):
    """ops.wgrad_split_grouped: conv (implicit, im2col view on the reduction side) and plain weight gradients with few output tiles and
    long reductions, one grouped + K-split launch and one grouped fold per kind; against the single-launch path (which is pinned
    against F.conv2d autograd above) and against fp32 torch for the plain ones."""
    torch.manual_seed(5)
    items, want = [], []
    for (B, H, Cc, Co, stride) in ((8, 56, 96, 192, 2), (8, 28, 192, 384, 2), (8, 56, 96, 96, 1), (4, 28, 192, 192, 1)):
        Ho = (H + 2 - 3) // stride + 1
        x = rnd(B * H * H, Cc, scale=0.5, seed=H + Cc)
        dy = rnd(B * Ho * Ho, Co, scale=0.3, seed=H + Co + 1)
        Kp = 9 * Cc
        items.append((dy, x, Co, Kp, dy.shape[0], (B, H, H, Cc, 3, stride)))
        want.append(ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True, M=Co, N=Kp, K=dy.shape[0], conv=(B, H, H, Cc, 3, stride)))
    for (M, N, K) in ((96, 32, 50176), (96, 32, 25088), (48, 64, 12544)):
        dy, x = rnd(K, M, scale=0.3, seed=K % 97), rnd(K, N, scale=0.3, seed=K % 89)
        items.append((dy, x, M, N, K, None))
        want.append(dy.float().t() @ x.float())
    from prismer_amd import _lib
    import ctypes
    cnt = (ctypes.c_int64 * 16)()
    _lib.lib.ph_gemm_dispatch_counts(cnt, 16, 1)
    got = ops.wgrad_split_grouped(items)
    torch.cuda.synchronize()
    _lib.lib.ph_gemm_dispatch_counts(cnt, 16, 0)
    assert cnt[5] == 2 and cnt[6] == 2, list(cnt)[:7]            # two grouped GEMM launches, two grouped fold passes
    for w, g, it in zip(want, got, items):
        assert rel_fro(g, w) < 3e-4, (it[2:5], rel_fro(g, w))


def test_gemm_deferred_fold_passes_match_immediate(ops):
    """defer_reduce: the fold passes of several split-K GEMMs are queued and run as ONE grouped launch per flush (both fold forms:
    one thread per output vector, 16 lanes per vector for many splits of a small output); results bit-identical to the immediate
    path (same partial sums, same summation order), the queue survives a workspace wrap-around, an unsplit call is unaffected."""
    cases = [(768, 768, 4160, 7), (96, 32, 100352, 64), (192, 864, 25088, 0), (96, 32, 100352, 256), (384, 1728, 6272, 0)]
    ops_in = [(rnd(K, M, scale=0.3, seed=20 + i), rnd(K, N, scale=0.3, seed=40 + i)) for i, (M, N, K, _) in enumerate(cases)]
    want = [ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True, split_k=sp) for (dy, x), (_, _, _, sp) in zip(ops_in, cases)]
    got = [ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True, split_k=sp, defer_reduce=True) for (dy, x), (_, _, _, sp) in zip(ops_in, cases)]
    small = ops.gemm(rnd(64, 64, seed=1), rnd(64, 64, seed=2), defer_reduce=True)          # no split: nothing queued for it
    ops.gemm_flush_deferred()
    ops.gemm_flush_deferred()                                                              # empty queue: no-op
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        assert torch.equal(w, g)
    assert rel_fro(small, rnd(64, 64, seed=1).float() @ rnd(64, 64, seed=2).float().t()) < 6e-3
    # more partial sums than the workspace holds: the queue flushes itself in between
    M, N, K = 768, 3456, 6272
    dy, x = rnd(K, M, scale=0.3, seed=77), rnd(K, N, scale=0.3, seed=78)
    ref = ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True, split_k=12)
    outs = [ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True, split_k=12, defer_reduce=True) for _ in range(4)]   # 4 x 127 MB > 256 MB
    ops.gemm_flush_deferred()
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)


@pytest.mark.parametrize('shapes', [
    [(768, 768, 960), (2304, 768, 960), (384, 768, 960), (768, 3072, 960), (3072, 768, 960)],      # one decoder layer: 128x128 tiles
    [(384, 768, 1984), (768, 384, 1984), (100, 200, 1984)],                                        # few tiles: 64x64 path, ragged edge
    [(64, 72, 130)],                                                                               # K tail (K % 64 != 0)
])
def test_wgrad_queue_grouped_launch(ops, shapes):
    """deferred weight gradients: ph_gemm_grouped_bf16 / ph_colsum_grouped_bf16 == the one-by-one launches."""
    probs = []
    for i, (M, N, K) in enumerate(shapes):
        Mp, Np = (M + 7) // 8 * 8, (N + 7) // 8 * 8
        dy = torch.zeros(K, Mp, dtype=BF, device='cuda'); dy[:, :M] = rnd(K, M, scale=0.3, seed=100 + i)
        x = torch.zeros(K, Np, dtype=BF, device='cuda'); x[:, :N] = rnd(K, N, scale=0.3, seed=200 + i)
        gw0 = torch.randn(M, N, device='cuda')
        gb0 = torch.randn(M, device='cuda')
        probs.append((dy, x, gw0, gb0, M, N, K))
    q = ops.WgradQueue()
    outs = []
    for dy, x, gw0, gb0, M, N, K in probs:
        gw, gb = gw0.clone(), gb0.clone()
        q.add_gemm(dy, x, gw, M, N, K)
        q.add_colsum(dy, gb, M)
        outs.append((gw, gb))
    assert any(q.gemms.values()) and q.cols, 'nothing was deferred'
    q.flush()
    torch.cuda.synchronize()
    for (dy, x, gw0, gb0, M, N, K), (gw, gb) in zip(probs, outs):
        ref = dy[:, :M].float().t() @ x[:, :N].float()
        assert rel_fro(gw - gw0, ref) < 3e-4, (M, N, K, rel_fro(gw - gw0, ref))
        assert rel_fro(gb - gb0, dy[:, :M].float().sum(0)) < 1e-4
    # a second read-modify-write of the same output must not share a launch with the first
    dy, x, gw0, gb0, M, N, K = probs[0]
    gw = gw0.clone()
    q.add_gemm(dy, x, gw, M, N, K); q.add_gemm(dy, x, gw, M, N, K); q.flush()
    assert rel_fro(gw - gw0, 2 * (dy[:, :M].float().t() @ x[:, :N].float())) < 3e-4


def test_gemm_dropout_mask(ops):
    M, N, K = 96, 128, 64
    a = torch.eye(M, K, dtype=BF, device='cuda')                    # out = dropout(B^T rows) on the first 64 rows
    b = rnd(N, K, seed=13)
    seed = 0x1234567890ABCDEF
    d = ops.Dropout(0.1, seed_tensor(seed), 77)
    out = ops.gemm(a, b, drop=d)
    keep = dropout_keep_linear(M * N, seed, 77, 0.1).reshape(M, N)
    ref = (a.float() @ b.float().t()).cpu() * keep / 0.9
    assert rel_fro(out, ref) < 6e-3
    assert abs(keep.float().mean().item() - 0.9) < 0.02


def test_gemm_errors(ops):
    a, b = rnd(16, 12), rnd(16, 12)
    with pytest.raises(RuntimeError):
        ops.gemm(a, b)                  # lda = 12: not a multiple of 8


# ---------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize('M,D', [(37, 256), (960, 768), (515, 1024)])
def test_layernorm(ops, M, D):
    x = rnd(M, D, seed=20)
    g = 1 + 0.1 * rnd(D, dtype=torch.float32, seed=21)
    b = 0.1 * rnd(D, dtype=torch.float32, seed=22)
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    xf = x.float().requires_grad_(True)
    gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (D,), gf, bf, 1e-5)
    assert rel_fro(y, ref) < 6e-3
    assert rel_fro(mean, xf.mean(1)) < 1e-5
    dy = rnd(M, D, seed=23)
    dsk = rnd(M, D, seed=24)
    dg = torch.zeros(D, device='cuda'); db = torch.zeros(D, device='cuda')
    dx, _ = ops.layernorm_bwd(dy, x, mean, rstd, g, dskip=dsk, dgamma=dg, dbeta=db)
    ref.backward(dy.float())
    assert rel_fro(dx, xf.grad + dsk.float()) < 6e-3
    assert ops.WQ.lns, 'the gamma / beta reduction should have been deferred'
    ops.join_side()                                       # folds the deferred per-block partials (one grouped launch)
    assert rel_fro(dg, gf.grad) < 1e-4 and rel_fro(db, bf.grad) < 1e-4
    # immediate (non-deferred) path: in-kernel reduction
    ops.WQ.enabled = False
    try:
        dg2 = torch.zeros(D, device='cuda'); db2 = torch.zeros(D, device='cuda')
        ops.layernorm_bwd(dy, x, mean, rstd, g, dskip=dsk, dgamma=dg2, dbeta=db2)
        assert rel_fro(dg2, gf.grad) < 1e-4 and rel_fro(db2, bf.grad) < 1e-4
    finally:
        ops.WQ.enabled = True


@pytest.mark.parametrize('M,D', [(8320, 768), (1027, 1024), (2049, 1280), (1200, 256)])
def test_layernorm_fwd_half_wave_rows(ops, M, D):
    """ln_fwd16_kernel (round 5: half a wave per bf16 row, 16-B vectors; rows of a multiple of 256 elements, M >= 1024): against fp32 torch and
    against the one-row-per-wave kernel on the same call (ph_layernorm_tuning(0)), with the mapped second output, the fp32 copy and the
    saved statistics; ragged M (rows beyond M in the last block)."""
    from prismer_amd import _lib
    from prismer_amd._lib import RowMap
    x = rnd(M, D, seed=20)
    g = 1 + 0.1 * rnd(D, dtype=torch.float32, seed=21)
    b = 0.1 * rnd(D, dtype=torch.float32, seed=22)

    def run():
        y2 = torch.zeros(M + 5 * ((M + 99) // 100), D, dtype=BF, device='cuda')
        yf = torch.empty(M, D, device='cuda')
        a = _lib.LayerNormFwdArgs(x.data_ptr(), g.data_ptr(), b.data_ptr(), 0, _lib.IDENT, y2.data_ptr(), RowMap(100, 105, 5), 0, 0, M, D, 1e-5, 0, yf.data_ptr())
        y = torch.empty(M, D, dtype=BF, device='cuda')
        stats = torch.empty(2, M, device='cuda')
        a.y, a.mean, a.rstd = y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr()
        import ctypes as C
        _lib.check(_lib.lib.ph_layernorm_fwd(C.byref(a), torch.cuda.current_stream().cuda_stream), 'ph_layernorm_fwd')
        return y, y2, yf, stats
    new = run()
    assert _lib.lib.ph_layernorm_tuning(0) == 1
    try:
        old = run()
    finally:
        _lib.lib.ph_layernorm_tuning(1)
    ref = F.layer_norm(x.float(), (D,), g, b, 1e-5)
    assert rel_fro(new[0], ref) < 6e-3 and rel_fro(new[2], ref) < 1e-5
    assert rel_fro(new[3][0], x.float().mean(1)) < 1e-5
    for a_, b_, tol in zip(new, old, (2e-3, 2e-3, 1e-5, 1e-5)):        # (bf16 outputs: a different summation order flips single roundings)
        assert rel_fro(a_, b_) < tol, rel_fro(a_, b_)
    rows = torch.arange(M, device='cuda')
    assert torch.equal(new[1][(rows // 100) * 105 + 5 + rows % 100], new[0])      # the mapped copy is the output, row for row


def test_layernorm_rowmap_and_dropout(ops):
    from prismer_amd._lib import RowMap
    B, L, Mx, D = 3, 8, 20, 256
    x = rnd(B * Mx, D, seed=25)
    g = torch.ones(D, device='cuda'); b = torch.zeros(D, device='cuda')
    kv = torch.zeros(B * (L + Mx), D, dtype=BF, device='cuda')
    ops.layernorm_fwd(x, g, b, out=kv, out_map=RowMap(Mx, L + Mx, L))
    ref = F.layer_norm(x.float(), (D,))
    got = kv.reshape(B, L + Mx, D)[:, L:].reshape(B * Mx, D)
    assert rel_fro(got, ref) < 6e-3 and kv.reshape(B, L + Mx, D)[:, :L].abs().max() == 0
    # backward reading dy through the same map + dropout-masked second output
    _, mean, rstd = ops.layernorm_fwd(x, g, b)
    dkv = rnd(B * (L + Mx), D, seed=26)
    seed = 99
    d = ops.Dropout(0.25, seed_tensor(seed), 5)
    dx, dxd = ops.layernorm_bwd(dkv, x, mean, rstd, g, dy_map=RowMap(Mx, L + Mx, L), drop=d)
    xf = x.float().requires_grad_(True)
    F.layer_norm(xf, (D,)).backward(dkv.float().reshape(B, L + Mx, D)[:, L:].reshape(B * Mx, D))
    assert rel_fro(dx, xf.grad) < 6e-3
    keep = dropout_keep_linear(B * Mx * D, seed, 5, 0.25).reshape(B * Mx, D)
    assert rel_fro(dxd, dx.float().cpu() * keep / 0.75) < 6e-3


# ---------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, scale, key_mask=None, causal=False, keep=None, p=0.0):
    """q [B,H,Sq,dh] fp32 etc.; finfo.min masking + clamp like roberta.py:113-115."""
    s = q @ k.transpose(-1, -2) * scale
    Sq, Sk = s.shape[-2:]
    m = torch.ones(Sq, Sk, dtype=torch.bool, device=q.device)
    if causal:
        m = torch.tril(m)
    m = m[None, None].expand_as(s).clone()
    if key_mask is not None:
        m &= key_mask.bool()[:, None, None, :]
    s = s.masked_fill(~m, torch.finfo(torch.float32).min)
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep.to(pr.device).reshape(pr.shape) / (1 - p)
    return pr @ v


ATTN_CASES = [  # B, H, Sq, Sk, dh, causal, masked, drop
    (2, 3, 260, 260, 64, False, False, 0.0),     # ViT self-attention (vit.py:53)
    (2, 8, 64, 300, 96, False, False, 0.0),      # perceiver cross-attention, head dim 96 (resampler.py:31)
    (3, 4, 30, 30, 64, True, True, 0.0),         # decoder self-attention: causal + padding (roberta.py:110-115)
    (2, 4, 30, 260, 64, False, False, 0.1),      # decoder cross-attention with prob-dropout (roberta.py:123)
    (2, 2, 70, 130, 32, True, False, 0.0),
    (1, 2, 100, 200, 128, False, True, 0.2),
    (2, 8, 64, 300, 160, False, False, 0.0),     # Prismer-HUGE resampler: ViT-H width 1280 / 8 heads (configs/prismer.json:50-73, resampler.py:18-24)
    # round 5: the small-query kernels (head dim 64, Sq <= 32, Sk <= 320: one block per (batch, head), keys split over the waves)
    (3, 4, 30, 30, 64, True, True, 0.1),         # decoder self-attention as trained: causal + padding + dropout
    (2, 12, 30, 260, 64, False, False, 0.1),     # decoder cross-attention at BASE geometry (9 key units over 4 waves)
    (2, 4, 40, 40, 64, True, True, 0.1),         # VQA text length (prismer_vqa.py:22-30): Sq > 32 -> the streaming kernels on both sides
    (2, 4, 30, 40, 64, True, True, 0.1),         # two key units over four waves (waves without a unit in the fused backward)
    (2, 4, 20, 80, 64, False, True, 0.1),        # three key units: the 3-unit forward merge (Sk 65..96)
    (2, 3, 1, 23, 64, True, True, 0.0),          # one query: a cached decode step (generation)
    (2, 2, 17, 320, 64, False, True, 0.2),       # the largest key image the small kernels take
    (1, 2, 64, 37, 64, False, False, 0.0),       # full 64-query tile, ragged keys (streaming kernels: Sq > 32)
    (2, 3, 280, 150, 64, False, False, 0.0),     # mask-free kernels, ragged tails on both sides
    # round 6: the head-resident kernels (plain, head dim 64, Sq, Sk <= 272: one block per (batch, head) or per share of its sub-tiles, NT = 13 or 17)
    (2, 4, 196, 196, 64, False, False, 0.0),     # PrismerZ-BASE: rgb tokens only (NT = 13, 12.25 tiles)
    (1, 3, 100, 272, 64, False, False, 0.0),     # the longest key sequence it takes (17 full tiles)
    (1, 2, 130, 209, 64, False, False, 0.0),     # first length on the NT = 17 instantiation, 14 tiles used, odd tail
    (2, 2, 50, 16, 64, False, False, 0.0),       # one key tile
    (40, 8, 260, 260, 64, False, False, 0.0),    # 320 heads: one block per head (split = 1), waves walking 5 / 4 / 4 / 4 sub-tiles
    (3, 2, 272, 33, 64, False, False, 0.0),      # 17 query sub-tiles against 3 key tiles
]


@pytest.mark.parametrize('B,H,Sq,Sk,dh,causal,masked,p', ATTN_CASES)
def test_attention(ops, B, H, Sq, Sk, dh, causal, masked, p):
    D = H * dh
    q = rnd(B * Sq, D, seed=30)
    kv = rnd(B * Sk, 2 * D, seed=31)             # packed [k | v] rows like the KV projection output
    k, v = kv[:, :D], kv[:, D:]
    km = None
    if masked:
        km = torch.ones(B, Sk, dtype=torch.uint8, device='cuda')
        for b in range(B):
            km[b, Sk - 3 - 2 * b:] = 0
    seed = 4242
    drop = ops.Dropout(p, seed_tensor(seed), 11) if p > 0 else None
    qs, ks = (Sq * D, D), (Sk * 2 * D, 2 * D)
    o, lse = ops.attention_fwd(q, k, v, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, key_mask=km, causal=causal,
                               drop=drop)
    qf = q.float().reshape(B, Sq, H, dh).transpose(1, 2).requires_grad_(True)
    kf = k.float().reshape(B, Sk, H, dh).transpose(1, 2).requires_grad_(True)
    vf = v.float().reshape(B, Sk, H, dh).transpose(1, 2).requires_grad_(True)
    keep = dropout_keep_attention(B * H, Sq, Sk, seed, 11, p).reshape(B, H, Sq, Sk).cuda() if p > 0 else None
    ref = attn_ref(qf, kf, vf, dh ** -0.5, km, causal, keep, p)
    ref_flat = ref.transpose(1, 2).reshape(B * Sq, D)
    assert rel_fro(o, ref_flat) < 8e-3, rel_fro(o, ref_flat)
    d_o = rnd(B * Sq, D, seed=32)
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    ops.attention_bwd(d_o, q, k, v, o, lse, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, dq=dq, dk=dkv[:, :D],
                      dv=dkv[:, D:], dq_strides=qs, dk_strides=ks, dv_strides=ks, key_mask=km, causal=causal, drop=drop)
    ref.backward(d_o.float().reshape(B, Sq, H, dh).transpose(1, 2))
    gq = qf.grad.transpose(1, 2).reshape(B * Sq, D)
    gk = kf.grad.transpose(1, 2).reshape(B * Sk, D)
    gv = vf.grad.transpose(1, 2).reshape(B * Sk, D)
    assert rel_fro(dq, gq) < 1.5e-2, ('dq', rel_fro(dq, gq))
    assert rel_fro(dkv[:, :D], gk) < 1.5e-2, ('dk', rel_fro(dkv[:, :D], gk))
    assert rel_fro(dkv[:, D:], gv) < 1.5e-2, ('dv', rel_fro(dkv[:, D:], gv))
    if dh == 64 and 32 < Sq <= 272 and Sk <= 272 and not (causal or masked or p > 0):
        # head-resident kernels (default, ran above) against the streaming kernels on the same launch: same P / dP / delta arithmetic
        from prismer_amd import _lib
        assert _lib.lib.ph_attention_tuning(2) == 1
        try:
            o2, lse2 = ops.attention_fwd(q, k, v, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, key_mask=km, causal=causal, drop=drop)
            dq2 = torch.empty_like(q); dkv2 = torch.empty_like(kv)
            ops.attention_bwd(d_o, q, k, v, o2, lse2, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, dq=dq2, dk=dkv2[:, :D],
                              dv=dkv2[:, D:], dq_strides=qs, dk_strides=ks, dv_strides=ks, key_mask=km, causal=causal, drop=drop)
        finally:
            assert _lib.lib.ph_attention_tuning(1) == 2
        assert rel_fro(o, o2) < 4e-3 and rel_fro(lse, lse2) < 1e-5, (rel_fro(o, o2), rel_fro(lse, lse2))
        assert rel_fro(dq, dq2) < 4e-3 and rel_fro(dkv, dkv2) < 4e-3, (rel_fro(dq, dq2), rel_fro(dkv, dkv2))
    if dh == 64 and Sq <= 32 and Sk <= 320:
        # both kernel families on the same launch: the small-query kernels ran above (default); the streaming kernels must agree with them
        # far inside the reference tolerance (same recomputed P / dP, same dropout words, different summation order)
        from prismer_amd import _lib
        assert _lib.lib.ph_attention_tuning(0) == 1
        try:
            o2, lse2 = ops.attention_fwd(q, k, v, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, key_mask=km, causal=causal, drop=drop)
            dq2 = torch.empty_like(q); dkv2 = torch.empty_like(kv)
            ops.attention_bwd(d_o, q, k, v, o2, lse2, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, dq=dq2, dk=dkv2[:, :D],
                              dv=dkv2[:, D:], dq_strides=qs, dk_strides=ks, dv_strides=ks, key_mask=km, causal=causal, drop=drop)
        finally:
            _lib.lib.ph_attention_tuning(1)
        assert rel_fro(o, o2) < 4e-3 and rel_fro(lse, lse2) < 1e-5, (rel_fro(o, o2), rel_fro(lse, lse2))
        assert rel_fro(dq, dq2) < 6e-3 and rel_fro(dkv, dkv2) < 6e-3, (rel_fro(dq, dq2), rel_fro(dkv, dkv2))


# ---------------------------------------------------------------------------------------------- front end
def test_patchify_and_resize(ops):
    img = torch.randn(2, 3, 64, 64, device='cuda')
    col = ops.patchify(img, 16, 768)
    ref = F.unfold(img, 16, stride=16).transpose(1, 2).reshape(-1, 3, 256).transpose(1, 2).reshape(-1, 768)   # (py, px, c) order
    assert rel_fro(col, ref) < 4e-3
    col14 = ops.patchify(torch.randn(1, 3, 56, 56, device='cuda'), 14, 592)
    assert col14.shape == (16, 592) and col14[:, 588:].abs().max() == 0
    x = torch.randn(2, 64, 48, 48, device='cuda')
    y = ops.resize_to_nhwc(x, 12, 12)
    ref = F.interpolate(x, size=(12, 12), mode='bilinear', align_corners=True).permute(0, 2, 3, 1)
    assert rel_fro(y, ref) < 4e-3
    x3 = torch.randn(2, 3, 28, 28, device='cuda')
    y3 = ops.resize_to_nhwc(x3, 32, 32)                                       # p=14 style up-scaling, C % 8 != 0
    assert rel_fro(y3, F.interpolate(x3, size=(32, 32), mode='bilinear', align_corners=True).permute(0, 2, 3, 1)) < 4e-3
    y1 = ops.resize_to_nhwc(x3[:, :1].contiguous(), 28, 28)                   # identity resize = layout change
    assert rel_fro(y1, x3[:, :1].permute(0, 2, 3, 1)) < 4e-3


@pytest.mark.parametrize('C,stride,ks', [(64, 2, 3), (32, 1, 3), (3, 2, 3), (1, 2, 3), (64, 1, 1)])
def test_conv_as_gemm(ops, C, stride, ks):
    B, H, W, Co = 2, 20, 20, 48
    x = rnd(B, H, W, C, seed=40)                                              # NHWC
    w = rnd(Co, C, ks, ks, scale=0.2, seed=41, dtype=torch.float32)
    K = ks * ks * C
    Kp = (K + 7) // 8 * 8
    shadow = torch.empty(Co, Kp, dtype=BF, device='cuda')
    ops.conv_weight_to_shadow(w, shadow, Co, C, ks, Kp)
    sc = 1 + 0.1 * rnd(C, dtype=torch.float32, seed=42)
    sh = 0.1 * rnd(C, dtype=torch.float32, seed=43)
    for bn in (False, True):
        col = ops.im2col(x, B, H, W, C, ks, stride, Kp, sc if bn else None, sh if bn else None)
        y = ops.gemm(col, shadow)
        xin = x.float().permute(0, 3, 1, 2)
        if bn:
            xin = torch.relu(xin * sc[None, :, None, None] + sh[None, :, None, None])
        ref = F.conv2d(bf16_round(xin), bf16_round(w), stride=stride, padding=ks // 2).permute(0, 2, 3, 1).reshape(-1, Co)
        assert rel_fro(y, ref) < 8e-3, (bn, rel_fro(y, ref))
    if C % 8 == 0:
        Ho = ops.conv_out_size(H, ks, stride)
        dcol = rnd(B * Ho * Ho, Kp, seed=44)
        dx = ops.col2im(dcol, B, H, W, C, ks, stride, Kp)
        xr = torch.zeros(B, C, H, W, device='cuda', requires_grad=True)
        cols = F.unfold(xr, ks, padding=ks // 2, stride=stride)               # [B, C*ks*ks, L] with (c, ky, kx) order
        d = dcol.float()[:, :K].reshape(B, Ho * Ho, ks * ks, C).permute(0, 3, 2, 1).reshape(B, C * ks * ks, Ho * Ho)
        cols.backward(d)
        assert rel_fro(dx.reshape(B, H, W, C), xr.grad.permute(0, 2, 3, 1)) < 6e-3
    # weight-gradient layout round trip
    ds = torch.randn(Co, Kp, device='cuda')
    dw = torch.zeros_like(w)
    ops.conv_grad_from_shadow(ds, dw, Co, C, ks, Kp)
    assert rel_fro(dw, ds[:, :K].reshape(Co, ks, ks, C).permute(0, 3, 1, 2)) < 1e-6


def test_conv_layout_grouped_kernels(ops):
    """the per-step grouped layout passes of the stems (weights -> bf16 im2col-order shadows, shadow-order gradients -> parameter
    layout), one block per output channel through LDS: against the single-layer kernels / plain permutes, for 3x3, 1x1 and the
    16x16 patch convolution, ragged K padding included"""
    layers = [(96, 64, 3), (192, 96, 3), (40, 3, 3), (24, 1, 3), (128, 128, 1), (64, 3, 16), (768, 384, 3)]
    items_w, items_g, refs = [], [], []
    for i, (Co, Ci, ks) in enumerate(layers):
        K = ks * ks * Ci
        Kp = (K + 7) // 8 * 8
        w = rnd(Co, Ci, ks, ks, scale=0.3, seed=60 + i, dtype=torch.float32)
        sh = torch.full((Co, Kp), 7.0, dtype=BF, device='cuda')
        ds = torch.randn(Co, Kp, device='cuda')
        dw0 = torch.randn(Co, Ci, ks, ks, device='cuda')
        dw = dw0.clone()
        items_w.append((w, sh, Co, Ci, ks, Kp)); items_g.append((ds, dw, Co, Ci, ks, Kp))
        refs.append((w, ds, dw0, K, Kp))
    ops.conv_layout_grouped(items_w, True)
    ops.conv_layout_grouped(items_g, False)
    for (w, sh, Co, Ci, ks, Kp), (ds, dw, _, _, _, _), (w_, ds_, dw0, K, _) in zip(items_w, items_g, refs):
        want = w.permute(0, 2, 3, 1).reshape(Co, K).to(BF)
        assert torch.equal(sh[:, :K], want) and float(sh[:, K:].float().abs().sum()) == 0.0, (Co, Ci, ks)
        want_g = dw0 + ds[:, :K].reshape(Co, ks, ks, Ci).permute(0, 3, 1, 2)
        assert torch.equal(dw, want_g), (Co, Ci, ks)


@pytest.mark.parametrize('B,H,C,Cout,stride,ks', [(3, 28, 96, 192, 2, 3), (2, 14, 64, 96, 1, 3), (2, 15, 384, 200, 2, 3), (5, 7, 768, 768, 1, 1),
                                                  (32, 56, 96, 192, 2, 3), (4, 15, 384, 200, 2, 3), (3, 20, 40, 72, 1, 3), (8, 28, 384, 768, 2, 3)])
def test_implicit_gemm_conv_forward_stats_and_wgrad(ops, B, H, C, Cout, stride, ks):
    """conv as implicit GEMM (no im2col matrix): forward y = conv(x, w) with BatchNorm statistics from the epilogue, and the weight
    gradient dW = dY^T . im2col(x) with the gathered operand on the reduction side -- against F.conv2d / autograd in fp32."""
    x = rnd(B, H, H, C, scale=0.7, seed=31)                                   # NHWC
    w = (torch.randn(Cout, C, ks, ks, generator=torch.Generator().manual_seed(7)) * 0.05).cuda()
    Kp = (ks * ks * C + 7) // 8 * 8
    shadow = torch.zeros(Cout, Kp, dtype=BF, device='cuda')
    ops.conv_weight_to_shadow(w, shadow, Cout, C, ks, Kp)
    Ho = ops.conv_out_size(H, ks, stride)
    M = B * Ho * Ho
    geo = (B, H, H, C, ks, stride)
    y = torch.empty(M, Cout, dtype=BF, device='cuda')
    slabs = torch.zeros(8, 2, Cout, device='cuda', dtype=torch.float64)          # PH_COLSTAT_SLABS replicated accumulators
    import ctypes
    from prismer_amd import _lib
    cnt = (ctypes.c_int64 * 16)()
    _lib.lib.ph_gemm_dispatch_counts(cnt, 16, 1)
    ops.conv_fwd_grouped([(x, geo, shadow, y, slabs)])
    torch.cuda.synchronize()
    _lib.lib.ph_gemm_dispatch_counts(cnt, 16, 0)
    # round 4: >= 256 rows and K >= 128 take the 256x128 LDS-DMA kernel (gather in the DMA source address), the rest the register-staged one
    assert (cnt[4], cnt[5]) == ((1, 0) if M >= 256 and Kp >= 128 else (0, 1)), list(cnt)[:7]
    stats = slabs.sum(0)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    wr = shadow[:, :ks * ks * C].float().view(Cout, ks, ks, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=stride, padding=ks // 2)
    ref2 = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    assert rel_fro(y, ref2) < 6e-3
    yf = y.float()
    assert rel_fro(stats[0], yf.double().sum(0)) < 1e-5 and rel_fro(stats[1], (yf.double() ** 2).sum(0)) < 1e-5      # statistics of the ROUNDED outputs
    # same result through the single-problem entry (routes to the grouped kernel) and against the explicit im2col path
    col = ops.im2col(x, B, H, H, C, ks, stride, Kp)
    y2 = ops.gemm(col, shadow)
    assert torch.equal(y, y2) or rel_fro(y, y2.float()) < 1e-3
    y3 = ops.gemm(x.view(-1, C), shadow, M=M, N=Cout, K=Kp, conv=geo)
    assert torch.equal(y3, y)
    # weight gradient
    dy = rnd(M, Cout, scale=0.3, seed=33)
    ds = ops.gemm(dy, x.view(-1, C), trans_a=True, trans_b=True, out_f32=True, M=Cout, N=Kp, K=M, conv=geo)
    ds_ref = ops.gemm(dy, col, trans_a=True, trans_b=True, out_f32=True, M=Cout, N=Kp, K=M)
    assert rel_fro(ds, ds_ref) < 2e-4
    ref.backward(dy.float().view(B, Ho, Ho, Cout).permute(0, 3, 1, 2))
    dw_ref = wr.grad.permute(0, 2, 3, 1).reshape(Cout, ks * ks * C)
    assert rel_fro(ds[:, :ks * ks * C], dw_ref) < 3e-4
    assert float(ds[:, ks * ks * C:].abs().sum()) == 0.0


@pytest.mark.parametrize('B,H,C,Cout,stride', [(2, 16, 24, 40, 2), (3, 12, 96, 64, 1), (2, 28, 96, 192, 2), (1, 8, 8, 16, 2), (4, 56, 96, 192, 2),
                                               (8, 28, 200, 384, 2), (5, 14, 384, 768, 1)])
def test_implicit_conv_data_gradient(ops, B, H, C, Cout, stride):
    """round 3: dX of a 3x3 pad-1 convolution as implicit GEMMs gathered from dY (stride 1: one flipped-tap convolution; stride 2: one
    problem per parity class of the input pixel, rows scattered through the output row map) -- against F.conv2d autograd in fp32 and
    against the dcol = dY.W + col2im path it replaces."""
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(Cout, C, 3, 3, generator=g) * 0.05).cuda()
    Ho = ops.conv_out_size(H, 3, stride)
    M = B * Ho * Ho
    dy = rnd(M, Cout, scale=0.3, seed=34)
    wd = torch.zeros(C, 9 * Cout, dtype=BF, device='cuda')
    ops.conv_dgrad_shadows([(w, wd, Cout, C, stride)])
    dx = torch.full((B * H * H, C), float('nan'), dtype=BF, device='cuda')          # every row must be written exactly once
    ops.conv_dgrad_grouped([(dy, wd, dx, (B, H, H, C, Cout, stride))])
    wb = w.to(BF).float()                                                            # the bf16-rounded weights the kernel multiplies
    xr = torch.zeros(B, C, H, H, device='cuda', requires_grad=True)
    F.conv2d(xr, wb, stride=stride, padding=1).backward(dy.float().view(B, Ho, Ho, Cout).permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1).reshape(B * H * H, C)
    assert torch.isfinite(dx.float()).all()
    assert rel_fro(dx, ref) < 6e-3, rel_fro(dx, ref)
    # the round-2 path: dcol = dY . W (shadow layout) then the col2im gather
    Kp = (9 * C + 7) // 8 * 8
    shadow = torch.zeros(Cout, Kp, dtype=BF, device='cuda')
    ops.conv_weight_to_shadow(w, shadow, Cout, C, 3, Kp)
    dcol = ops.gemm(dy, shadow, trans_b=True)
    old = ops.col2im(dcol, B, H, H, C, 3, stride, Kp)
    assert rel_fro(dx, old.float().view(B * H * H, C)) < 1e-2                        # (dcol is rounded to bf16 before the 9-tap sum)


@pytest.mark.parametrize('M,C', [(5000, 96), (777, 32), (3000, 768)])
def test_batchnorm(ops, M, C):
    y = rnd(M, C, seed=50) * 2 + 0.5
    g = 1 + 0.1 * rnd(C, dtype=torch.float32, seed=51)
    b = 0.1 * rnd(C, dtype=torch.float32, seed=52)
    rm = 0.1 * rnd(C, dtype=torch.float32, seed=53); rv = 1 + 0.2 * rnd(C, dtype=torch.float32, seed=54).abs()
    rm0, rv0 = rm.clone(), rv.clone()
    st = ops.bn_stats(y, g, b, rm, rv, True)
    yf = y.float().requires_grad_(True)
    gf, bf_ = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rmr, rvr = rm0.clone(), rv0.clone()
    ref = torch.relu(F.batch_norm(yf, rmr, rvr, gf, bf_, True, 0.1, 1e-5))
    assert rel_fro(st[0], yf.mean(0)) < 1e-4 and rel_fro(st[1], (yf.var(0, unbiased=False) + 1e-5).rsqrt()) < 1e-3
    assert rel_fro(rm, rmr) < 1e-4 and rel_fro(rv, rvr) < 1e-3
    got = torch.relu(y.float() * st[2] + st[3])
    assert rel_fro(got, ref) < 2e-3
    da = rnd(M, C, seed=55)
    dg = torch.zeros(C, device='cuda'); db = torch.zeros(C, device='cuda')
    dy = ops.bn_relu_bwd(da, y, g, b, st, dg, db)
    ref.backward(da.float())
    assert rel_fro(dy, yf.grad) < 1e-2, rel_fro(dy, yf.grad)
    assert rel_fro(dg, gf.grad) < 2e-3 and rel_fro(db, bf_.grad) < 2e-3
    ev = ops.bn_stats(y, g, b, rm, rv, False)
    assert rel_fro(ev[2], g * (rv + 1e-5).rsqrt()) < 1e-5


@pytest.mark.parametrize('shapes', [[(5000, 96), (3001, 768)], [(777, 160), (1030, 1280), (641, 2048), (4099, 32)]])
def test_batchnorm_grouped(ops, shapes):
    """ph_bn_apply_relu_grouped / ph_bn_relu_bwd_grouped (the stems' same-index layers in one launch; vit.py:88-120 BatchNorm2d + ReLU in train
    mode and their autograd) against torch: channel counts whose 8-channel groups do not divide the block (96, 160, 768, 1280), one row per
    block iteration (2048), ragged row counts.  The forward takes its statistics from fp64 column sums as the conv epilogue leaves them."""
    fwd, bwd, refs = [], [], []
    for k, (M, C) in enumerate(shapes):
        y = rnd(M, C, seed=60 + k) * 2 + 0.5
        g = 1 + 0.1 * rnd(C, dtype=torch.float32, seed=70 + k)
        b = 0.1 * rnd(C, dtype=torch.float32, seed=80 + k)
        rm = 0.1 * rnd(C, dtype=torch.float32, seed=90 + k); rv = 1 + 0.2 * rnd(C, dtype=torch.float32, seed=100 + k).abs()
        rmr, rvr = rm.clone(), rv.clone()
        sums = torch.zeros(8, 2, C, dtype=torch.float64, device='cuda')
        sums[k % 8, 0] = y.double().sum(0); sums[(k + 3) % 8, 1] = (y.double() ** 2).sum(0)        # (any split over the slabs)
        a, stats = torch.empty_like(y), torch.zeros(4, C, dtype=torch.float32, device='cuda')
        fwd.append(dict(y=y, a=a, gamma=g, beta=b, running_mean=rm, running_var=rv, stats=stats, sums=sums))
        yf = y.float().requires_grad_(True)
        gf, bf_ = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = torch.relu(F.batch_norm(yf, rmr, rvr, gf, bf_, True, 0.1, 1e-5))
        da = rnd(M, C, seed=110 + k)
        ref.backward(da.float())
        refs.append((ref, yf, gf, bf_, rmr, rvr))
        bwd.append(dict(y=y, a=da, dy=torch.empty_like(y), gamma=g, beta=b, stats=stats, sums=torch.zeros(2 * C, dtype=torch.float32, device='cuda'),
                        dgamma=torch.zeros(C, device='cuda'), dbeta=torch.zeros(C, device='cuda')))
    ops.bn_apply_relu_grouped(fwd, True)
    ops.bn_relu_bwd_grouped(bwd)
    for f, bw, (ref, yf, gf, bf_, rmr, rvr) in zip(fwd, bwd, refs):
        assert rel_fro(f['a'], ref) < 5e-3, rel_fro(f['a'], ref)                                     # (bf16 output)
        assert rel_fro(f['running_mean'], rmr) < 1e-4 and rel_fro(f['running_var'], rvr) < 1e-3
        assert rel_fro(bw['dy'], yf.grad) < 1e-2, rel_fro(bw['dy'], yf.grad)
        assert rel_fro(bw['dgamma'], gf.grad) < 3e-3 and rel_fro(bw['dbeta'], bf_.grad) < 3e-3
    # eval mode: running statistics, no update
    rm0 = fwd[0]['running_mean'].clone()
    ops.bn_apply_relu_grouped(fwd[:1], False)
    y, g, b = fwd[0]['y'].float(), fwd[0]['gamma'], fwd[0]['beta']
    want = torch.relu((y - fwd[0]['running_mean']) * (fwd[0]['running_var'] + 1e-5).rsqrt() * g + b)
    assert rel_fro(fwd[0]['a'], want) < 5e-3 and torch.equal(fwd[0]['running_mean'], rm0)


@pytest.mark.parametrize('B', [2, 19])          # 19: batch slices (gridDim.y = 4) with a ragged last slice
def test_tokens_finalize(ops, B):
    g_, D, E = 4, 256, 64
    G = g_ * g_
    feat = rnd(B * G, D, seed=60)
    pos = rnd(G, D, dtype=torch.float32, seed=61)
    inst = torch.randint(0, 256, (B, 1, E, E), device='cuda')
    table = torch.randint(0, 128, (256,), dtype=torch.int32, device='cuda')
    emb = rnd(128, D, dtype=torch.float32, seed=62)
    tpb, off = 3 * G, G
    tok = torch.zeros(B * tpb, D, dtype=BF, device='cuda')
    ops.tokens_finalize(feat, pos, tok, B, G, D, tpb, off, inst, E, g_, table, emb)
    im = F.interpolate(inst.float(), size=(g_, g_), mode='nearest')[:, 0].long()            # vit.py:142
    ref = feat.float().reshape(B, G, D) + pos[None] + emb[table.long()[im]].reshape(B, G, D)
    assert rel_fro(tok.reshape(B, tpb, D)[:, off:off + G], ref) < 4e-3
    dtok = rnd(B * tpb, D, seed=63)
    dfeat = torch.empty(B * G, D, dtype=BF, device='cuda')
    dpos = torch.zeros(G, D, device='cuda'); demb = torch.zeros(128, D, device='cuda')
    ops.tokens_finalize_bwd(dtok, dfeat, dpos, B, G, D, tpb, off, inst, E, g_, table, demb)
    sl = dtok.float().reshape(B, tpb, D)[:, off:off + G]
    assert torch.equal(dfeat.reshape(B, G, D), dtok.reshape(B, tpb, D)[:, off:off + G])
    assert rel_fro(dpos, sl.sum(0)) < 1e-5
    ref_e = torch.zeros(128, D, device='cuda').index_add_(0, table.long()[im].reshape(-1), sl.reshape(-1, D))
    assert rel_fro(demb, ref_e) < 1e-5


def test_taps(ops):
    n_in, n_out, taps, D = 36, 16, 16, 64
    inp = torch.randn(n_in, D, device='cuda')
    idx = torch.randint(0, n_in, (n_out, taps), dtype=torch.int32, device='cuda')
    w = torch.randn(n_out, taps, device='cuda')
    out = ops.gather_taps(inp, idx, w, n_out, taps, D)
    ref = (w[:, :, None] * inp[idx.long()]).sum(1)
    assert rel_fro(out, ref) < 1e-5
    din = torch.zeros(n_in, D, device='cuda')
    dout = torch.randn(n_out, D, device='cuda')
    ops.scatter_taps(dout, din, idx, w, n_out, taps, D)
    ref_in = torch.zeros(n_in, D, device='cuda').index_add_(0, idx.long().reshape(-1), (w[:, :, None] * dout[:, None]).reshape(-1, D))
    assert rel_fro(din, ref_in) < 1e-5


# ---------------------------------------------------------------------------------------------- decoder ends
def test_embed(ops):
    B, T, H, V, pad = 3, 12, 256, 1003, 1
    ids = torch.randint(3, V, (B, T), device='cuda')
    ids[:, 0] = 0; ids[1, 8:] = pad; ids[2, 5:] = pad
    word = rnd(V, H, scale=0.05, dtype=torch.float32, seed=70); posw = rnd(514, H, scale=0.05, dtype=torch.float32, seed=71)
    typ = rnd(1, H, scale=0.05, dtype=torch.float32, seed=72)
    g = 1 + 0.1 * rnd(H, dtype=torch.float32, seed=73); b = 0.1 * rnd(H, dtype=torch.float32, seed=74)
    seed = 31337
    d = ops.Dropout(0.1, seed_tensor(seed), 3)
    out, xhat, rstd = ops.embed_fwd(ids, word, posw, typ, g, b, 1e-5, pad, d)
    m = (ids != pad).long()
    pid = torch.cumsum(m, 1) * m + pad
    wf, pf, tf = word.clone().requires_grad_(True), posw.clone().requires_grad_(True), typ.clone().requires_grad_(True)
    gf, bf_ = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    e = F.embedding(ids, wf, padding_idx=pad) + tf[0] + F.embedding(pid, pf, padding_idx=pad)
    keep = dropout_keep_linear(B * T * H, seed, 3, 0.1).reshape(B * T, H).cuda()
    ref = F.layer_norm(e, (H,), gf, bf_, 1e-5).reshape(B * T, H) * keep / 0.9
    assert rel_fro(out, ref) < 6e-3
    dout = rnd(B * T, H, seed=75)
    dw, dp, dt = torch.zeros_like(word), torch.zeros_like(posw), torch.zeros_like(typ)
    dg, db = torch.zeros_like(g), torch.zeros_like(b)
    ops.embed_bwd(dout, ids, word, posw, typ, g, b, 1e-5, pad, xhat, rstd, d, dw, dp, dt, dg, db)
    ref.backward(dout.float())
    for got, want, name in ((dw, wf.grad, 'word'), (dp, pf.grad, 'pos'), (dt, tf.grad, 'type'), (dg, gf.grad, 'gamma'), (db, bf_.grad, 'beta')):
        assert rel_fro(got, want) < 1e-2, (name, rel_fro(got, want))
    assert dw[pad].abs().max() == 0 and dp[pad].abs().max() == 0


@pytest.mark.parametrize('V', [1003, 50265])
def test_cross_entropy(ops, V):
    B, T = 3, 9
    Vp = (V + 63) // 64 * 64
    logits = torch.zeros(B * T, Vp, dtype=BF, device='cuda')
    logits[:, :V] = rnd(B * T, V, scale=2.0, seed=80)
    logits[:, V:] = 7.0                                        # garbage in the pad columns must be ignored
    labels = torch.randint(0, V, (B, T), device='cuda')
    labels[:, :3] = -100; labels[1, 6:] = -100
    loss, lse = ops.ce_fwd(logits, labels, B, T, V, 0.1)
    lf = logits[:, :V].float().reshape(B, T, V).requires_grad_(True)
    ref = F.cross_entropy(lf[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), reduction='none', label_smoothing=0.1).view(B, -1).sum(1)
    assert rel_fro(loss, ref) < 1e-4, (loss, ref)
    dl = torch.tensor([0.5, 0.25, 1.0], device='cuda')
    (ref * dl).sum().backward()
    ops.ce_bwd(logits, labels, B, T, V, 0.1, lse, dl)
    assert rel_fro(logits[:, :V].reshape(B, T, V), lf.grad) < 8e-3
    assert logits[:, V:].abs().max() == 0


def test_cross_entropy_under_graph_replay_needs_no_zeroed_buffers(ops):
    """Round 5: ph_ce_fwd used to zero its accumulator with hipMemsetAsync; captured, that is a 128-byte memset NODE, and such a node does
    not replay correctly on ROCm 7.0 (tools/graph_memset_probe.py) -- the replayed step's loss was garbage while everything else was
    healthy (round-4 loader leg).  Now: per-token losses + a fixed-order sum, every output word overwritten.  Captured with the B = 32
    geometry of the benchmark, replayed over buffers that hold junk, against new logits each replay."""
    B, T, V = 32, 30, 1003
    Vp = (V + 63) // 64 * 64
    logits = torch.zeros(B * T, Vp, dtype=BF, device='cuda')
    labels = torch.randint(0, V, (B, T), device='cuda')
    labels[:, :4] = -100
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.ce_fwd(logits, labels, B, T, V, 0.1)                          # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        junk = torch.empty(B, device='cuda')                              # the block the loss vector will take
        junk.fill_(-1.0e30)
        del junk
        loss, lse = ops.ce_fwd(logits, labels, B, T, V, 0.1)
        loss.mul_(1.0)                                                    # a kernel node behind it
    for i in range(6):
        logits[:, :V] = rnd(B * T, V, scale=2.0, seed=90 + i)
        g.replay()
        torch.cuda.synchronize()
        ref = F.cross_entropy(logits[:, :V].float().reshape(B, T, V)[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), reduction='none',
                              label_smoothing=0.1).view(B, -1).sum(1)
        assert rel_fro(loss, ref) < 1e-4, (i, loss[:4], ref[:4])
    # the sum is a fixed-order reduction: bit-identical from run to run (the atomics it replaced were not)
    first = loss.clone()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(first, loss)


# ---------------------------------------------------------------------------------------------- optimizer / utils
def test_adamw_matches_torch(ops):
    n = 10007
    p = torch.randn(n, device='cuda')
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=5e-5, weight_decay=0.05)
    m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda'); pb = torch.empty(n, dtype=BF, device='cuda')
    for step in (1, 2, 3):
        g = torch.randn(n, device='cuda')
        ref_p.grad = g.clone()
        opt.step()
        hyper = torch.tensor([5e-5, 1 - 0.9 ** step, 1 - 0.999 ** step], device='cuda')
        ops.adamw(p, g, m, v, pb, n, hyper, zero_grad=(step == 3))
        assert (g.abs().sum() == 0) == (step == 3)            # zero_grad: the gradient buffer is cleared after it was read
    assert max_abs(p, ref_p) < 2.5e-6                          # parameters of magnitude ~1: a few fp32 ulps over three steps
    assert rel_fro(pb, p) < 4e-3


def test_fill_zero_and_copy_kernels(ops):
    """ops.zeros / ops.copy_flat: the step's accumulator resets and the optimizer-sharding staging copies run as library kernels (round 5: no
    memset / memcpy node and no stock torch kernel inside a captured segment), also under hipGraph replay over buffers that held junk."""
    z = ops.zeros((7, 13), torch.float32, 'cuda')
    assert z.shape == (7, 13) and z.dtype == torch.float32 and float(z.abs().sum()) == 0.0
    z64 = ops.zeros(5, torch.float64, 'cuda')
    assert z64.dtype == torch.float64 and float(z64.abs().sum()) == 0.0
    src = torch.randn(100003, device='cuda')
    for n in (100003, 4096, 3):
        dst = torch.full((n,), 7.0, device='cuda')
        ops.copy_flat(dst, src[:n])
        assert torch.equal(dst, src[:n])
    off = torch.full((1001,), 7.0, device='cuda')
    ops.copy_flat(off[1:], src[:1000])                           # misaligned destination: falls back to Tensor.copy_
    assert torch.equal(off[1:], src[:1000]) and off[0] == 7.0
    buf = torch.full((1024,), 9.0, device='cuda')
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        t = ops.zeros(32, torch.float32, 'cuda')                 # a 128-byte reset: the size a memset node gets wrong
        t.add_(buf[:32])
        ops.copy_flat(buf[512:544], t)
    for i in range(4):
        buf[:32] = float(i + 1)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(buf[512:544], torch.full((32,), float(i + 1), device='cuda')), i


def test_small_utils(ops):
    x = torch.randn(1000, 72, device='cuda')
    xb = ops.cast_to_bf16(x)
    assert torch.equal(xb, x.to(BF))
    assert torch.equal(ops.cast_to_f32(xb), xb.float())
    out = torch.ones(72, device='cuda')
    ops.colsum(xb, out)
    assert rel_fro(out, 1 + xb.float().sum(0)) < 1e-5
    a, b = rnd(999, 8, seed=90), rnd(999, 8, seed=91)
    assert rel_fro(ops.add(a, b), a.float() + b.float()) < 4e-3
    af = a.float().requires_grad_(True)
    F.gelu(af).backward(b.float())
    assert rel_fro(ops.act_bwd(b, a, 3), af.grad) < 6e-3
    from prismer_amd._lib import RowMap
    src = rnd(12, 64, seed=92)
    dst = torch.zeros(3 * 10, 64, dtype=BF, device='cuda')
    ops.copy_rows(src, dst, 12, 64, dst_map=RowMap(4, 10, 2))
    assert torch.equal(dst.reshape(3, 10, 64)[:, 2:6].reshape(12, 64), src)
    s = torch.tensor([5], dtype=torch.int64, device='cuda')
    ops.advance_seed(s)
    assert s.item() != 5


def test_softmax_gather(ops):
    """ph_softmax_gather_bf16 (round 6: the first-token candidate probabilities of inference='rank', prismer_caption.py:70) against torch on a
    strided view of a padded logits buffer (the decoder's [B, T, Vpad] layout), vocabulary not a multiple of 8, out-of-range ids -> 0"""
    B, T, V, Vp = 5, 7, 50265, 50304
    buf = (torch.randn(B, T, Vp, device='cuda') * 3).to(BF)
    last = buf[:, -1, :V]
    ids = torch.tensor([0, 5, 50264, 1234, 5, 49999], device='cuda')
    got = ops.softmax_gather(last, ids)
    ref = torch.softmax(last.float(), dim=1).index_select(1, ids)
    assert got.shape == (B, 6) and rel_fro(got, ref) < 1e-5
    assert ops.softmax_gather(last, torch.tensor([V + 3], device='cuda')).abs().max().item() == 0.0



def test_store_words_carries_host_scalars_in_kernel_arguments(ops):
    """ph_store_words: the per-step learning rate / bias corrections (fp32) and the 256 instance draws (int32) reach device memory through the
    arguments of one launch; stream-ordered (a later call overwrites), bounds checked"""
    import random as _r
    hyper = torch.full((4,), -1.0, dtype=torch.float32, device='cuda')
    table = torch.full((260,), -7, dtype=torch.int32, device='cuda')
    vals = (5e-5 * 0.37, 1.0 - 0.9 ** 17, 1.0 - 0.999 ** 17)
    draws = [_r.randint(0, 127) for _ in range(256)]
    ops.store_words(hyper, vals, table, draws)
    assert torch.equal(hyper.cpu(), torch.tensor(list(vals) + [-1.0], dtype=torch.float32))          # (same double -> fp32 rounding as torch.tensor)
    assert table[:256].cpu().tolist() == draws and table[256:].cpu().tolist() == [-7] * 4
    ops.store_words(hyper, (1.0, 2.0, 3.0))
    assert hyper.cpu().tolist() == [1.0, 2.0, 3.0, -1.0] and table[:256].cpu().tolist() == draws
    with pytest.raises(RuntimeError):
        ops.store_words(torch.zeros(400, dtype=torch.float32, device='cuda'), [0.0] * 321)
