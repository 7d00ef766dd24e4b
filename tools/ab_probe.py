"""A/B kernel probe: times the same launches against two (or more) builds of libprismer_hip.so inside ONE process,
interleaved (A, B, A, B ...) under hipGraph replay, so clock / thermal drift cannot masquerade as a kernel change.

    AB_LIBS=prismer_amd/lib/libprismer_hip_old.so,prismer_amd/lib/libprismer_hip.so python tools/ab_probe.py [gemm|attn|ln ...]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import _lib, ops
from prismer_amd._lib import ACT_GELU, ACT_QUICKGELU

BF = torch.bfloat16


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


LIBS = [(os.path.basename(p), load(p)) for p in os.environ.get('AB_LIBS', _lib.LIB_PATH).split(',')]


def use(lib):
    ops.lib = lib
    _lib.lib = lib


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


def ab(name, fn, n=20, rounds=4, flops=None):
    graphs = []
    for _, lib in LIBS:
        use(lib)
        graphs.append(graph_of(fn, n))
    best = [1e30] * len(LIBS)
    for _ in range(rounds):
        for i, g in enumerate(graphs):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                g.replay()
            b.record(); torch.cuda.synchronize()
            best[i] = min(best[i], a.elapsed_time(b) / (3 * n) * 1e3)
    msg = ' | '.join(f'{LIBS[i][0][-12:]}: {best[i]:7.1f} us' + (f' {flops / best[i] / 1e6:5.0f} TF' if flops else '') for i in range(len(LIBS)))
    print(f'{name:34s} {msg}', flush=True)
    use(LIBS[-1][1])


def gemm_case(name, M, N, K, layout='nt', **kw):
    dev = 'cuda'
    if layout == 'nt':
        a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF); lk = {}
    elif layout == 'nn':
        a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(K, N, device=dev).to(BF); lk = dict(trans_b=True)
    else:
        a = torch.randn(K, M, device=dev).to(BF); b = torch.randn(K, N, device=dev).to(BF); lk = dict(trans_a=True, trans_b=True)
    args = {}
    if kw.get('bias'):
        args['bias'] = torch.randn(N, device=dev)
    if kw.get('act'):
        args['act'] = kw['act']
    if kw.get('pre_out'):
        args['pre_out'] = torch.empty(M, N, dtype=BF, device=dev)
    if kw.get('act_in'):
        args['act_in'] = torch.randn(M, N, device=dev).to(BF)
    if kw.get('residual') == 'bf16':
        args['residual'] = torch.randn(M, N, device=dev).to(BF)
    elif kw.get('residual') == 'f32':
        args['residual'] = torch.randn(M, N, device=dev)
    if kw.get('drop'):
        args['drop'] = ops.Dropout(0.1, torch.tensor([1234], dtype=torch.int64, device=dev), 7)
    f32 = kw.get('out_f32', False)
    if f32:
        args.update(out_f32=True, accumulate=True)
    out = torch.zeros(M, N, dtype=torch.float32 if f32 else BF, device=dev)
    ab(f'gemm {name} {layout} {M}x{N}x{K}', lambda: ops.gemm(a, b, out=out, **lk, **args), flops=2.0 * M * N * K)


def epi_suite():
    gemm_case('plain', 8320, 3072, 768)
    gemm_case('bias', 8320, 3072, 768, bias=True)
    gemm_case('bias+qgelu', 8320, 3072, 768, bias=True, act=ACT_QUICKGELU)
    gemm_case('bias+pre', 8320, 3072, 768, bias=True, pre_out=True)
    gemm_case('bias+qgelu+pre', 8320, 3072, 768, bias=True, act=ACT_QUICKGELU, pre_out=True)
    gemm_case('res bf16', 8320, 3072, 768, residual='bf16')
    gemm_case('act_in', 8320, 3072, 768, act=ACT_QUICKGELU, act_in=True)
    gemm_case('f32 out acc', 8320, 3072, 768, out_f32=True)


def gemm_suite():
    gemm_case('plain', 8320, 3072, 768)
    gemm_case('c_fc b+qgelu+pre', 8320, 3072, 768, bias=True, act=ACT_QUICKGELU, pre_out=True)
    gemm_case('dgrad act_in', 8320, 3072, 768, 'nn', act=ACT_QUICKGELU, act_in=True)
    gemm_case('qkv bias', 8320, 2304, 768, bias=True)
    gemm_case('out b+res', 8320, 768, 768, bias=True, residual='bf16')
    gemm_case('c_proj b+res', 8320, 768, 3072, bias=True, residual='bf16')
    gemm_case('dgrad qkv', 8320, 768, 2304, 'nn')
    gemm_case('wgrad 768', 768, 768, 8320, 'tn', out_f32=True)
    gemm_case('wgrad fc', 3072, 768, 8320, 'tn', out_f32=True)
    gemm_case('dec dense b+drop+rf32', 960, 768, 768, bias=True, drop=True, residual='f32')
    gemm_case('dec up b+gelu+pre', 960, 3072, 768, bias=True, act=ACT_GELU, pre_out=True)
    gemm_case('dec down b+drop+rf32', 960, 768, 3072, bias=True, drop=True, residual='f32')
    gemm_case('dec wgrad', 768, 768, 960, 'tn', out_f32=True)
    gemm_case('lm head bias', 960, 50304, 768, bias=True)
    gemm_case('resampler kv', 39680, 1536, 768, bias=True)


def attn_case(name, B, H, Sq, Sk, dh, causal=False, mask=False, drop=False):
    dev = 'cuda'
    W = H * dh
    q = torch.randn(B * Sq, W, device=dev).to(BF); k = torch.randn(B * Sk, W, device=dev).to(BF); v = torch.randn(B * Sk, W, device=dev).to(BF)
    do = torch.randn(B * Sq, W, device=dev).to(BF)
    km = None
    if mask:
        km = torch.ones(B, Sk, dtype=torch.uint8, device=dev); km[:, Sk - 5:] = 0
    dr = ops.Dropout(0.1, torch.tensor([77], dtype=torch.int64, device=dev), 3) if drop else None
    qs, ks = (Sq * W, W), (Sk * W, W)
    o, lse = ops.attention_fwd(q, k, v, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, key_mask=km, causal=causal, drop=dr)
    fl = 4.0 * B * H * Sq * Sk * dh
    ab(f'attn fwd {name}', lambda: ops.attention_fwd(q, k, v, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks, key_mask=km,
                                                      causal=causal, drop=dr, out=o), flops=fl)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ab(f'attn bwd {name}', lambda: ops.attention_bwd(do, q, k, v, o, lse, B, H, Sq, Sk, dh, q_strides=qs, k_strides=ks, v_strides=ks,
                                                      dq=dq, dk=dk, dv=dv, dq_strides=qs, dk_strides=ks, dv_strides=ks, key_mask=km,
                                                      causal=causal, drop=dr), flops=2.5 * fl)


def attn_suite():
    attn_case('vit S=260 dh64', 32, 12, 260, 260, 64)
    attn_case('resampler 64x1304 dh96', 32, 8, 64, 1304, 96)
    attn_case('dec self T=30 causal+mask+drop', 32, 12, 30, 30, 64, causal=True, mask=True, drop=True)
    attn_case('dec cross 30x260 drop', 32, 12, 30, 260, 64, drop=True)
    attn_case('large vit S=1240 dh64', 4, 16, 1240, 1240, 64)


def ln_suite():
    dev = 'cuda'
    for M, D, f32 in ((8320, 768, False), (960, 768, True), (39680, 768, False), (8320, 1024, False)):
        x = torch.randn(M, D, device=dev)
        xx = x if f32 else x.to(BF)
        g, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
        y, mean, rstd = ops.layernorm_fwd(xx, g, b)
        ab(f'ln fwd M={M} D={D} f32={int(f32)}', lambda: ops.layernorm_fwd(xx, g, b, out=y))
        dy = torch.randn(M, D, device=dev).to(BF)
        dsk = torch.randn(M, D, device=dev).to(BF)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        dx = torch.empty(M, D, dtype=BF, device=dev)
        ab(f'ln bwd M={M} D={D} f32={int(f32)} +dskip', lambda: ops.layernorm_bwd(dy, xx, mean, rstd, g, dskip=dsk, dgamma=dg, dbeta=db, dx=dx))


def front_suite():
    """stem byte movers at the dense experts' layer-2 / layer-3 geometry (bs32)."""
    dev = 'cuda'
    import ctypes as Ct
    from prismer_amd._lib import check
    for (B, H, Cc, s_) in ((32, 112, 96, 2), (32, 56, 192, 2), (32, 28, 384, 2)):
        Kp = 9 * Cc
        x = torch.randn(B * H * H, Cc, device=dev).to(BF)
        sc, sh = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
        Ho = (H + 2 - 3) // s_ + 1
        col = torch.empty(B * Ho * Ho, Kp, dtype=BF, device=dev)
        dx = torch.empty(B * H * H, Cc, dtype=BF, device=dev)
        ab(f'im2col {H}x{H}x{Cc} s{s_}', lambda: check(ops.lib.ph_im2col_nhwc(x.data_ptr(), col.data_ptr(), B, H, H, Cc, 3, s_, Kp, sc.data_ptr(),
                                                                         sh.data_ptr(), ops._stream()), 'im2col'))
        ab(f'col2im {H}x{H}x{Cc} s{s_}', lambda: check(ops.lib.ph_col2im_nhwc(col.data_ptr(), dx.data_ptr(), B, H, H, Cc, 3, s_, Kp, ops._stream()),
                                                      'col2im'))
        M = B * Ho * Ho
        C2 = 2 * Cc
        y = torch.randn(M, C2, device=dev).to(BF); da = torch.randn(M, C2, device=dev).to(BF)
        g, b_ = torch.rand(C2, device=dev) + 0.5, torch.randn(C2, device=dev) * 0.1
        rm, rv = torch.zeros(C2, device=dev), torch.ones(C2, device=dev)
        st = ops.bn_stats(y, g, b_, rm, rv, True)
        dg, db = torch.zeros(C2, device=dev), torch.zeros(C2, device=dev)
        ab(f'bn_stats M={M} C={C2}', lambda: ops.bn_stats(y, g, b_, rm, rv, True))
        ab(f'bn_relu_bwd M={M} C={C2}', lambda: ops.bn_relu_bwd(da, y, g, b_, st, dg, db))


def tokens_suite():
    dev = 'cuda'
    B, G, D, tpb, off = 32, 196, 768, 1176, 392
    dtok = torch.randn(B * tpb, D, device=dev).to(BF)
    dfeat = torch.empty(B * G, D, dtype=BF, device=dev)
    dpos = torch.zeros(G, D, device=dev)
    ab('tokens_bwd 32x196x768 (no instance)', lambda: ops.tokens_finalize_bwd(dtok, dfeat, dpos, B, G, D, tpb, off))
    inst = torch.randint(0, 12, (B, 1, 224, 224), device=dev)
    table = torch.randint(0, 128, (256,), dtype=torch.int32, device=dev)
    demb = torch.zeros(128, D, device=dev)
    ab('tokens_bwd 32x196x768 (instance)', lambda: ops.tokens_finalize_bwd(dtok, dfeat, dpos, B, G, D, tpb, off, inst, 224, 14, table, demb))


if __name__ == '__main__':
    which = sys.argv[1:] or ['gemm', 'attn', 'ln']
    print('libs:', [n for n, _ in LIBS])
    if 'attn' in which:
        attn_suite()
    if 'ln' in which:
        ln_suite()
    if 'gemm' in which:
        gemm_suite()
    if 'epi' in which:
        epi_suite()
    if 'front' in which:
        front_suite()
    if 'tokens' in which:
        tokens_suite()
