"""Where does a GEMM launch spend its time?  s_memtime stamps from the -DPH_TIMELINE diagnostics build (never the product library):

    python tools/build_variant.py tl gemm_s64.hip:-DPH_TIMELINE gemm_big.hip:-DPH_TIMELINE [attention.hip:-DPH_TIMELINE norm.hip:-DPH_TIMELINE]
    PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_tl.so python tools/timeline_probe.py

Per block and thread group (waves 0 / 4 of the 256x128 kernel, thread groups 0 / 1 of the k-split kernel) the kernels keep
    0 entry | 1 prologue loads issued | 2 first tile visible | 3 k loop done | 4 ring drained (big kernel only) | 5 tile parked in LDS
    6 write-out: first group's input loads issued | 7 first group's stores issued | 8 write-out done | 9 stores acknowledged (vmcnt 0)
Printed per case: the launch duration by HIP events, the tick rate implied by it, and the median / p90 / max over blocks of every
segment in microseconds, plus the spread of block start and end times (dispatch skew, tail)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from prismer_amd import _lib, ops
from prismer_amd._lib import ACT_QUICKGELU, ACT_GELU

BF = torch.bfloat16
SLOTS, BLOCKS = 16, 4096
lib = _lib.lib
for f in ('ph_tl_fetch_big', 'ph_tl_fetch_gemm', 'ph_tl_fetch_attn', 'ph_tl_fetch_norm'):
    if hasattr(lib, f):                                  # (only the translation units built with -DPH_TIMELINE export theirs)
        getattr(lib, f).restype = C.c_int
        getattr(lib, f).argtypes = [C.c_void_p, C.c_int, C.c_int]

FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device='cuda')     # larger than L2 + the 256 MB Infinity Cache


def fetch(which):
    buf = np.zeros(BLOCKS * 2 * SLOTS, dtype=np.uint64)
    getattr(lib, which)(buf.ctypes.data, buf.size, 1)
    return buf.reshape(BLOCKS, 2, SLOTS)


SEG_SMALL = [('entry->loads issued', 0, 1), ('->first data', 1, 2), ('->body done', 2, 3), ('->stores issued', 3, 8), ('store ack', 8, 9)]
SEG = [('entry->issued', 0, 1), ('issued->tile0', 1, 2), ('k loop', 2, 3), ('drain', 3, 4), ('park', 4, 5), ('wo loads', 5, 6),
       ('wo group0', 6, 7), ('wo rest', 7, 8), ('store ack', 8, 9)]


def report(name, tl, us, nblk, flops, SEG=SEG):
    """s_memtime is a per-XCD counter (the eight XCDs are ~1e9 ticks apart): durations inside a block are exact, start / end spreads are
    taken per XCD (slot 15 carries HW_REG_XCC_ID) and the worst XCD is printed."""
    t = tl[:nblk].astype(np.float64)
    xcc = (tl[:nblk, :, 15] >> np.uint64(32)).astype(np.int64) & 0xF
    live = t[:, :, 0] > 0
    lv = t[live]
    rt = (lv[:, 11] - lv[:, 10]) / 100.0                   # s_memrealtime: 100 MHz
    ok = rt > 0.5
    tick = float(np.median((lv[ok, 9] - lv[ok, 0]) / rt[ok])) if ok.any() else 2100.0      # s_memtime ticks per microsecond
    spans, skews = [], []
    for x in range(16):
        sel = live & (xcc == x)
        if sel.any():
            spans.append((t[:, :, 9][sel].max() - t[:, :, 0][sel].min()) / tick)
            skews.append((t[:, :, 0][sel].max() - t[:, :, 0][sel].min()) / tick)
    print(f'== {name}: {us:7.2f} us by events (single eager launch), {nblk} blocks, {tick:.0f} ticks/us; per XCD first-start -> last-end '
          f'{max(spans):.2f} us (min {min(spans):.2f}), start skew {max(skews):.2f} us; {flops / max(spans) / 1e6:.0f} TF over that span', flush=True)
    for g in (0, 1):
        if not live[:, g].any():
            continue
        tt = t[live[:, g], g, :]
        tot = (tt[:, 9] - tt[:, 0]) / tick
        parts = [f'in-block total {np.median(tot):5.2f}/{np.percentile(tot, 90):5.2f}/{tot.max():5.2f}']
        for label, a, b in SEG:
            if (tt[:, b] == 0).all() or (tt[:, a] == 0).all():
                continue
            d = (tt[:, b] - tt[:, a]) / tick
            parts.append(f'{label} {np.median(d):5.2f}/{np.percentile(d, 90):5.2f}/{d.max():5.2f}')
        print(f'   grp{g} med/p90/max us: ' + ' | '.join(parts), flush=True)


def run(name, which, M, N, K, cold, **kw):
    a = (torch.randn(M, K, device='cuda') * 0.5).to(BF)
    tb = kw.pop('tb', False)
    b = (torch.randn(K, N, device='cuda') * 0.05).to(BF) if tb else (torch.randn(N, K, device='cuda') * 0.05).to(BF)
    args = {}
    if tb:
        args['trans_b'] = True
    if kw.get('bias', True):
        args['bias'] = torch.randn(N, device='cuda') * 0.1
    if kw.get('act'):
        args['act'] = kw['act']
    if kw.get('residual'):
        args['residual'] = torch.randn(M, N, device='cuda').to(torch.float32 if kw.get('f32res') else BF)
    if kw.get('act_in'):
        args['act_in'] = torch.randn(M, N, device='cuda').to(BF)
        args['act'] = _lib.ACT_SAVED_GRAD
    out = torch.empty(M, N, dtype=torch.float32 if kw.get('out_f32') else BF, device='cuda')
    if kw.get('pre'):
        args['pre_out'] = torch.empty(M, N, dtype=BF, device='cuda')
        args['pre_grad'] = True
    if kw.get('drop'):
        args['drop'] = ops.Dropout(0.1, torch.zeros(1, dtype=torch.int64, device='cuda'), 7)
    call = lambda: ops.gemm(a, b, out=out, out_f32=bool(kw.get('out_f32')), **args)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    fetch(which)
    if cold:
        FLUSH.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    tl = fetch(which)
    nblk = int((tl[:, 0, 0] > 0).sum())
    report(f'{name} [{"cold" if cold else "warm"}]', tl, us, nblk, 2.0 * M * N * K)


def timed(name, which, call, flops, cold, seg=SEG_SMALL):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    fetch(which)
    if cold:
        FLUSH.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record()
    torch.cuda.synchronize()
    tl = fetch(which)
    nblk = int((tl[:, 0, 0] > 0).sum())
    report(f'{name} [{"cold" if cold else "warm"}]', tl, e0.elapsed_time(e1) * 1e3, nblk, flops, seg)


def attn_case(name, B, H, Sq, Sk, dh, cold, causal=False, masked=False, drop=False):
    W = H * dh
    q = torch.randn(B * Sq, W, device='cuda').to(BF)
    kv = torch.randn(B * Sk, 2 * W, device='cuda').to(BF)
    km = torch.ones(B, Sk, dtype=torch.uint8, device='cuda') if masked else None
    dr = ops.Dropout(0.1, torch.zeros(1, dtype=torch.int64, device='cuda'), 3) if drop else None
    ks = (Sk * 2 * W, 2 * W)
    call = lambda: ops.attention_fwd(q, kv[:, :W], kv[:, W:], B, H, Sq, Sk, dh, q_strides=(Sq * W, W), k_strides=ks, v_strides=ks,
                                     key_mask=km, causal=causal, drop=dr)
    timed(name, 'ph_tl_fetch_attn', call, 4.0 * B * H * Sq * Sk * dh, cold)


def ln_case(name, M, D, cold, f32):
    x = torch.randn(M, D, device='cuda')
    x = x if f32 else x.to(BF)
    g, b = torch.ones(D, device='cuda'), torch.zeros(D, device='cuda')
    yf = torch.empty(M, D, device='cuda') if f32 else None
    y = torch.empty(M, D, dtype=BF, device='cuda')
    call = lambda: ops.layernorm_fwd(x, g, b, out=y, out_f32=yf)
    timed(name, 'ph_tl_fetch_norm', call, 1.0, cold)


def chain_case(name, call, n=40):
    """per-launch time of n back-to-back identical launches under graph replay (boundary + kernel)"""
    call(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
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
    print(f'-- chain of {n}: {name}: {best:6.2f} us per launch', flush=True)


if __name__ == '__main__':
    torch.manual_seed(0)
    modes = [int(m) for m in os.environ.get('TL_BIG_MODES', '5,6').split(',')]       # 5: ping-pong, 6: + LEAN tail (default)
    for mode in modes:
        _lib.lib.ph_gemm_tuning(mode, 128)
        for cold in (False, True):
            tag = f'm{mode} '
            run(tag + 'big out-proj 8320x768x768 +bias +res', 'ph_tl_fetch_big', 8320, 768, 768, cold, residual=True)
            run(tag + 'big dgrad 8320x768x768 tb', 'ph_tl_fetch_big', 8320, 768, 768, cold, tb=True, bias=False)
            run(tag + 'big proj 8320x768x3072 +res', 'ph_tl_fetch_big', 8320, 768, 3072, cold, residual=True)
            run(tag + 'big qkv 8320x2304x768', 'ph_tl_fetch_big', 8320, 2304, 768, cold)
            run(tag + 'big c_fc 8192x3072x768 qgelu + grad', 'ph_tl_fetch_big', 8192, 3072, 768, cold, act=ACT_QUICKGELU, pre=True)
            run(tag + 'big dgrad c_proj 8192x3072x768 tb *saved', 'ph_tl_fetch_big', 8192, 3072, 768, cold, tb=True, bias=False, act_in=True)
            run(tag + 'big tiny-K 8320x768x128 +res (fixed cost alone)', 'ph_tl_fetch_big', 8320, 768, 128, cold, residual=True)
    _lib.lib.ph_gemm_tuning(*_lib.GEMM_BIG_DEFAULT)
    for cold in (False, True):
        run('ks2 dec dense 960x768x768 +bias +drop +res32', 'ph_tl_fetch_gemm', 960, 768, 768, cold, residual=True, f32res=True, drop=True)
        run('ks2 dec dgrad 960x768x768 tb', 'ph_tl_fetch_gemm', 960, 768, 768, cold, tb=True, bias=False)
        run('ks2 dec fc 960x3072x768 gelu + grad', 'ph_tl_fetch_gemm', 960, 3072, 768, cold, act=ACT_GELU, pre=True)
        run('ks2 dec proj 960x768x3072 +res32', 'ph_tl_fetch_gemm', 960, 768, 3072, cold, residual=True, f32res=True, drop=True)
        run('64 dec qkv 960x2304x768', 'ph_tl_fetch_gemm', 960, 2304, 768, cold)
        if os.environ.get('TL_SMALL', '1') != '0':
            attn_case('attn ViT 32x12 S=260 dh64 plain', 32, 12, 260, 260, 64, cold)
            attn_case('attn dec self 32x12 T=30 causal+mask+drop', 32, 12, 30, 30, 64, cold, causal=True, masked=True, drop=True)
            attn_case('attn dec cross 32x12 T=30 S=260 drop', 32, 12, 30, 260, 64, cold, drop=True)
            ln_case('ln dec 960x768 f32 in, bf16+f32 out', 960, 768, cold, True)
            ln_case('ln vit 8320x768 bf16', 8320, 768, cold, False)
    # per-launch cost in a dependent chain under graph replay (what the step pays): same launches, no stamps read
    x32 = torch.randn(960, 768, device='cuda'); g1, b1 = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
    y16, y32 = torch.empty(960, 768, dtype=BF, device='cuda'), torch.empty(960, 768, device='cuda')
    chain_case('ln dec 960x768', lambda: ops.layernorm_fwd(x32, g1, b1, out=y16, out_f32=y32))
    a = torch.randn(960, 768, device='cuda').to(BF); w = (torch.randn(768, 768, device='cuda') * 0.05).to(BF); bias = torch.zeros(768, device='cuda')
    o32 = torch.empty(960, 768, device='cuda')
    chain_case('ks2 960x768x768 +bias+res32 -> f32', lambda: ops.gemm(a, w, out=o32, out_f32=True, bias=bias, residual=x32))
    a8 = torch.randn(8320, 768, device='cuda').to(BF); r8 = torch.randn(8320, 768, device='cuda').to(BF); o8 = torch.empty(8320, 768, dtype=BF, device='cuda')
    a83 = torch.randn(8320, 3072, device='cuda').to(BF); w3 = (torch.randn(768, 3072, device='cuda') * 0.05).to(BF)
    wf = (torch.randn(3072, 768, device='cuda') * 0.05).to(BF); bf_ = torch.zeros(3072, device='cuda')
    of = torch.empty(8192, 3072, dtype=BF, device='cuda'); pf = torch.empty(8192, 3072, dtype=BF, device='cuda')
    wq = (torch.randn(2304, 768, device='cuda') * 0.05).to(BF); bq = torch.zeros(2304, device='cuda'); oq = torch.empty(8320, 2304, dtype=BF, device='cuda')
    for mode in modes:
        _lib.lib.ph_gemm_tuning(mode, 128)
        chain_case(f'm{mode} big 8320x768x768 +bias+res', lambda: ops.gemm(a8, w, out=o8, bias=bias, residual=r8))
        chain_case(f'm{mode} big 8320x768x768 plain', lambda: ops.gemm(a8, w, out=o8))
        chain_case(f'm{mode} big 8320x768x3072 +bias+res', lambda: ops.gemm(a83, w3, out=o8, bias=bias, residual=r8))
        chain_case(f'm{mode} big c_fc 8192x3072x768 qgelu+grad', lambda: ops.gemm(a8[:8192], wf, out=of, bias=bf_, act=ACT_QUICKGELU, pre_out=pf, pre_grad=True))
        chain_case(f'm{mode} big qkv 8320x2304x768 +bias', lambda: ops.gemm(a8, wq, out=oq, bias=bq))
    _lib.lib.ph_gemm_tuning(modes[0], 128)
    empty = torch.empty(64, device='cuda')
    chain_case('torch fill_ of 64 floats (boundary of a trivial kernel)', lambda: empty.fill_(1.0))
