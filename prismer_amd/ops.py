"""Tensor-level wrappers over the C ABI (prismer_amd/_lib.py).  torch supplies device memory and the current
HIP stream; every arithmetic step runs in libprismer_hip.so.  No fallbacks: a missing library fails at import.
"""
import ctypes as C
import os
import struct

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_RELU, ACT_RELU2, ACT_SAVED_GRAD, IDENT, RowMap, check, lib, ptr)

BF16 = torch.bfloat16
F32 = torch.float32


_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _stream():
    """the current HIP stream's handle (what torch.cuda.current_stream().cuda_stream returns, without building the Stream object: the
    eager drop-in step asks ~2000 times and is bound by host time, not by the GPU)"""
    return _raw_stream(_cur_device())


class Dropout:
    """Dropout descriptor: probability, device seed tensor (uint64 viewed as int64[1]) and a per-call-site stream id."""
    __slots__ = ('p', 'seed', 'stream')

    def __init__(self, p, seed, stream):
        self.p, self.seed, self.stream = float(p), seed, int(stream)


NO_DROP = None

_WS = {}
WS_BYTES = 256 << 20


def _workspace(device, which=0):
    """per-(device, stream) fp32 scratch for split-K partial tiles / LayerNorm partials: kernels on one stream are
    serialised, so one buffer per stream is race-free (allocated once, reused by every launch on that stream).
    which = 1: the separate buffer that holds the partial sums of DEFERRED fold passes until gemm_flush_deferred()"""
    key = (device, _raw_stream(device.index if device.index is not None else _cur_device()), which)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(WS_BYTES // 4, dtype=F32, device=device)
        _WS[key] = ws
    return ws


class SideStream:
    """Second HIP stream for work that is OFF the critical path of the backward pass (weight-gradient GEMMs, bias column
    sums): the dgrad chain keeps the main stream, the side stream fills the CUs its short / narrow launches leave idle.
    Under hipGraph capture the fork/join events become graph edges, i.e. parallel branches.  Tensors touched on the side
    stream are kept referenced until join() so the caching allocator cannot recycle them under the side stream's feet."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.keep = []
        self.dirty = False

    def run(self, fn, *tensors):
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            out = fn()
        self.keep.append(tensors)
        if out is not None:
            self.keep.append(out)
        self.dirty = True
        return out

    def wait(self):
        """current stream waits for everything queued on the side stream so far; keeps the tensor references (used inside
        micro-batch branches, where another branch's side work may still be pending)."""
        torch.cuda.current_stream().wait_stream(self.stream)

    def join(self):
        if self.dirty:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.dirty = False
        self.keep.clear()


class BranchPool:
    """N extra HIP streams for INDEPENDENT sub-graphs (the six expert stems): each branch forks from the current stream,
    runs on its own stream (parallel branches of the hipGraph) and join() makes the current stream wait for all of them.
    Tensors created inside a branch must stay referenced until join() (the callers keep them in their saved state)."""

    def __init__(self, device, n):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        self.used = set()

    def branch(self, i):
        s = self.streams[i % len(self.streams)]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        s.wait_event(ev)
        self.used.add(s)
        return torch.cuda.stream(s)

    def join(self):
        cur = torch.cuda.current_stream()
        for s in self.used:
            cur.wait_stream(s)
        self.used.clear()


class _NoPool:
    def branch(self, i):
        import contextlib
        return contextlib.nullcontext()

    def join(self):
        pass


POOL = _NoPool()     # set by the Trainer (native path)
MICRO = _NoPool()    # set by the Trainer: one stream per micro-batch (independent halves of the batch as parallel branches)
SIDE = None          # set by the Trainer (native path); None = everything on the current stream


def off_critical_path(fn, *tensors):
    """run fn() on the side stream when one is active, else inline."""
    if SIDE is None:
        return fn()
    return SIDE.run(fn, *tensors)


class WgradQueue:
    """Deferred weight gradients.  Nothing inside a step reads dW / db before the optimizer, so the wgrad GEMMs
    (dW += dY^T X: outputs of 18..144 tiles, alone they fill a fraction of the chip and need split-K + a reduce pass) and
    the bias column sums are queued and issued per reduction length K as ONE grouped launch each
    (ph_gemm_grouped_bf16 / ph_colsum_grouped_bf16).  join_side() flushes, so every existing synchronisation point of the
    programs still sees complete gradients.  Operand tensors stay referenced until their launch."""

    def __init__(self):
        self.enabled = True
        # False: nothing is launched before flush() -- the grouped launches are big (hundreds of 16 us tiles); issued while the
        # decoder's chain of small dependent kernels is running they take every CU slot and each chain kernel then waits for
        # slots to drain.  True: launch as soon as 16 problems of one reduction length are queued.
        self.eager_flush = False
        # > 0 (with eager_flush): mid-segment launches are BACKGROUND launches of at most this many blocks (each walks several
        # tiles), so the weight gradients trickle through the CUs the chain leaves idle; the flush at the segment end is uncapped
        self.bg_blocks = 0
        self.gemms = {}          # K -> [(dy, x, gw, M, N, K)]
        self.cols = []           # [(dy, gb, M, N)]
        self.folds = []          # [(dshadow, dw, Cout, Cin, ks, Kp)]  conv weight gradients: shadow layout -> parameter layout
        self.lns = []            # [(partials, blocks, D, dgamma, dbeta)]  LayerNorm gamma / beta partial sums
        self.ln_arena = None     # fp32 scratch the deferred LayerNorm backwards write their per-block partials into
        self.ln_off = 0
        self.ln_stream = None
        # first-writer bookkeeping of a native training step (Trainer._wq_scope): `counts` (while not None) tallies the writes per
        # output address of one whole step; `exclusive` = addresses written exactly once per step -> their GEMM overwrites
        self.counts = None
        self.exclusive = None

    def add_gemm(self, dy, x, gw, M, N, K):
        """gw[M, N] (fp32) += dy[K, M]^T . x[K, N]"""
        if self.counts is not None:
            key = gw.data_ptr()
            c = self.counts.get(key, (0, 0, True))
            self.counts[key] = (c[0] + 1, M * N, c[2] and gw.is_contiguous() and self.enabled)
        if not self.enabled:
            gemm(dy, x, out=gw, trans_a=True, trans_b=True, out_f32=True, accumulate=True, M=M, N=N, K=K)
            return
        q = self.gemms.setdefault(K, [])                           # (two writers of one output never share a launch: _flush_gemms)
        q.append((dy, x, gw, M, N, K))
        if self.eager_flush and len(q) == _lib.GEMM_GROUP_MAX:
            self._flush_gemms(K, self.bg_blocks)

    def add_colsum(self, dy, gb, N):
        if not self.enabled:
            colsum(dy, gb, N=N)
            return
        self.cols.append((dy, gb, dy.shape[0], N))
        if self.eager_flush and len(self.cols) == _lib.GEMM_GROUP_MAX:
            self._flush_cols()

    def add_conv_fold(self, ds, dw, Cout, Cin, ks, Kp):
        """dw[Cout, Cin, ks, ks] += ds[Cout, Kp] (im2col column order): queued behind the GEMM that produces ds on the
        side stream; all layers of a step fold in one launch"""
        if not self.enabled:
            conv_grad_from_shadow(ds, dw, Cout, Cin, ks, Kp)
            return
        self.folds.append((ds, dw, Cout, Cin, ks, Kp))

    LN_ARENA_FLOATS = 48 << 20          # 192 MB: one pass of the decoder (~80 MB) or the encoder (~110 MB) at bs32

    def ln_slot(self, blocks, D, dgamma, dbeta):
        """partials buffer [blocks, 2, D] for one LayerNorm backward whose gamma / beta reduction is deferred; None = do not
        defer (queue off, arena full, or a stream other than the one the pending partials were written on)"""
        if not self.enabled:
            return None
        cur = torch.cuda.current_stream().cuda_stream
        if self.lns and self.ln_stream != cur:
            return None
        n = blocks * 2 * D
        if self.ln_arena is None:
            self.ln_arena = torch.empty(self.LN_ARENA_FLOATS, dtype=F32, device=dgamma.device if dgamma is not None else dbeta.device)
        if self.ln_off + n > self.ln_arena.numel():
            self._flush_lns()           # same stream as the writers: the fold finishes before the arena is reused
        if n > self.ln_arena.numel():
            return None
        self.ln_stream = cur
        slot = self.ln_arena[self.ln_off:self.ln_off + n]
        self.ln_off += n
        self.lns.append((slot, blocks, D, dgamma, dbeta))
        return slot

    def _flush_lns(self):
        q, self.lns, self.ln_off = self.lns, [], 0
        for i0 in range(0, len(q), _lib.GEMM_GROUP_MAX):
            part = q[i0:i0 + _lib.GEMM_GROUP_MAX]
            arr = (_lib.LnReduceItem * len(part))()
            for it, (ws, blocks, D, dg, db) in zip(arr, part):
                it.ws, it.blocks, it.D, it.dgamma, it.dbeta = ws.data_ptr(), blocks, D, ptr(dg), ptr(db)
            check(lib.ph_ln_param_reduce_grouped(arr, len(part), _stream()), 'ph_ln_param_reduce_grouped')

    def _flush_gemms(self, K, max_blocks=0):
        allq = self.gemms.pop(K, [])
        if not allq:
            return

        excl = self.exclusive

        def work():
            i0 = 0
            while i0 < len(allq):                    # <= 16 problems per launch, no output twice in one launch
                q, seen = [], set()
                while i0 < len(allq) and len(q) < _lib.GEMM_GROUP_MAX and allq[i0][2].data_ptr() not in seen:
                    seen.add(allq[i0][2].data_ptr()); q.append(allq[i0]); i0 += 1
                arr = (_lib.GemmArgs * len(q))()
                for g, (dy, x, gw, M, N, Kk) in zip(arr, q):
                    g.A, g.B, g.C = dy.data_ptr(), x.data_ptr(), gw.data_ptr()
                    g.M, g.N, g.K = M, N, Kk
                    g.lda, g.ldb, g.ldc = dy.stride(0), x.stride(0), gw.stride(0)
                    # single-writer outputs of the native step are OVERWRITTEN (no read of the zeroed buffer, and AdamW does not
                    # have to zero it): self.exclusive is set by the Trainer for the duration of its own step only
                    acc = 0 if (excl is not None and gw.data_ptr() in excl) else 1
                    g.trans_a, g.trans_b, g.out_f32, g.accumulate, g.alpha = 1, 1, 1, acc, 1.0
                check(lib.ph_gemm_grouped_capped_bf16(arr, len(q), max_blocks, _stream()), 'ph_gemm_grouped_capped_bf16')
        off_critical_path(work, *[t for e in allq for t in e[:3]])

    def _flush_cols(self):
        allq, self.cols = self.cols, []
        if not allq:
            return

        def work():
            for i0 in range(0, len(allq), _lib.GEMM_GROUP_MAX):
                q = allq[i0:i0 + _lib.GEMM_GROUP_MAX]
                arr = (_lib.ColsumItem * len(q))()
                for it, (dy, gb, M, N) in zip(arr, q):
                    it.x, it.out, it.M, it.N, it.ld = dy.data_ptr(), gb.data_ptr(), M, N, dy.stride(0)
                check(lib.ph_colsum_grouped_bf16(arr, len(q), _stream()), 'ph_colsum_grouped_bf16')
        off_critical_path(work, *[t for e in allq for t in e[:2]])

    def flush(self):
        for K in list(self.gemms):
            self._flush_gemms(K)
        self._flush_cols()
        if self.lns:
            self._flush_lns()
        if self.folds:
            q, self.folds = self.folds, []
            off_critical_path(lambda: conv_layout_grouped(q, False), *[t for e in q for t in e[:2]])


WQ = WgradQueue()


def join_side():
    WQ.flush()
    if SIDE is not None:
        SIDE.join()


def wait_side():
    if SIDE is not None:
        SIDE.wait()


def gemm(a, b, out=None, *, trans_a=False, trans_b=False, bias=None, act=ACT_NONE, pre_out=None, act_in=None,
         residual=None, drop=None, out_f32=False, accumulate=False, alpha=1.0, split_k=0, M=None, N=None, K=None, pre_grad=False,
         conv=None, col_stats=None, defer_reduce=False):
    """out[M,N] = epi(alpha * opA . opB^T).  a: [M,K] (or [K,M] if trans_a), b: [N,K] (or [K,N] if trans_b)."""
    if M is None:
        M = a.shape[1] if trans_a else a.shape[0]
    if K is None:
        K = a.shape[0] if trans_a else a.shape[1]
    if N is None:
        N = b.shape[1] if trans_b else b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=F32 if out_f32 else BF16, device=a.device)
    g = _lib.GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = a.stride(0), b.stride(0), out.stride(0)
    g.trans_a, g.trans_b = int(trans_a), int(trans_b)
    cg = None
    if conv is not None:                              # (B, H, W, C, ks, stride): the A (forward) / B (wgrad) operand is an im2col view
        cg = _lib.ConvGather(*conv)
        g.conv = C.pointer(cg)
        if trans_a:
            g.ldb = N
        else:
            g.lda = K
    g.col_stats = ptr(col_stats)
    g.bias = ptr(bias)
    g.act = act
    g.pre_out = ptr(pre_out)
    g.pre_grad = int(pre_grad)
    g.act_in = ptr(act_in)
    g.ld_act = act_in.stride(0) if act_in is not None else 0
    g.residual = ptr(residual)
    g.residual_f32 = int(residual is not None and residual.dtype == F32)
    g.ldr = residual.stride(0) if residual is not None else 0
    if drop is not None and drop.p > 0.0:
        g.drop_p, g.drop_seed, g.drop_stream = drop.p, drop.seed.data_ptr(), drop.stream
    else:
        g.drop_p, g.drop_seed, g.drop_stream = 0.0, None, 0
    g.out_f32, g.accumulate, g.alpha, g.split_k = int(out_f32), int(accumulate), float(alpha), int(split_k)
    # defer_reduce: a split-K call leaves its fold pass to gemm_flush_deferred() (one grouped launch for up to 16 of them); the partial
    # sums of the queued calls live in their own workspace, which nothing else uses
    ws = _workspace(a.device, 1 if defer_reduce else 0)
    g.workspace, g.workspace_bytes, g.defer_reduce = ws.data_ptr(), WS_BYTES, int(defer_reduce)
    check(lib.ph_gemm_bf16(C.byref(g), _stream()), 'ph_gemm_bf16')
    return out


def gemm_flush_deferred():
    check(lib.ph_gemm_flush_deferred(_stream()), 'ph_gemm_flush_deferred')


def layernorm_fwd(x, gamma, beta, eps=1e-5, out=None, out_map=IDENT, out2=None, out2_map=IDENT, save_stats=True, out_f32=None):
    """x: bf16 or fp32 [M, D]; out: bf16; out_f32: optional fp32 [M, D] copy of the output (fp32 residual stream)."""
    M, D = x.shape
    if out is None:
        out = torch.empty((M, D), dtype=BF16, device=x.device)
    mean = rstd = None
    if save_stats:
        stats = torch.empty((2, M), dtype=F32, device=x.device)
        mean, rstd = stats[0], stats[1]
    a = _lib.LayerNormFwdArgs(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), out_map, ptr(out2), out2_map,
                              ptr(mean), ptr(rstd), M, D, eps, int(x.dtype == F32), ptr(out_f32))
    check(lib.ph_layernorm_fwd(C.byref(a), _stream()), 'ph_layernorm_fwd')
    return out, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, *, dy_map=IDENT, dy2=None, dy2_map=IDENT, dskip=None, dgamma=None, dbeta=None,
                  drop=None, dx=None, dx_drop=None):
    M, D = x.shape
    if dx is None:
        dx = torch.empty((M, D), dtype=BF16, device=x.device)
    a = _lib.LayerNormBwdArgs()
    a.x_f32 = int(x.dtype == F32)
    a.dy, a.dy_map, a.dy2, a.dy2_map = dy.data_ptr(), dy_map, ptr(dy2), dy2_map
    a.x, a.mean, a.rstd, a.gamma = x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr()
    a.dskip, a.dx = ptr(dskip), dx.data_ptr()
    if drop is not None and drop.p > 0.0:
        if dx_drop is None:
            dx_drop = torch.empty((M, D), dtype=BF16, device=x.device)
        a.dx_drop, a.drop_p, a.drop_seed, a.drop_stream = dx_drop.data_ptr(), drop.p, drop.seed.data_ptr(), drop.stream
    else:
        dx_drop = None
        a.dx_drop, a.drop_p, a.drop_seed, a.drop_stream = None, 0.0, None, 0
    a.dgamma, a.dbeta, a.M, a.D = ptr(dgamma), ptr(dbeta), M, D
    slot = None
    if dgamma is not None or dbeta is not None:
        slot = WQ.ln_slot(lib.ph_layernorm_bwd_blocks(M), D, dgamma, dbeta)     # gamma / beta reduction deferred and grouped
    if slot is not None:
        a.partial_ws, a.partial_ws_bytes, a.defer_reduce = slot.data_ptr(), slot.numel() * 4, 1
    else:
        ws = _workspace(x.device)
        a.partial_ws, a.partial_ws_bytes, a.defer_reduce = ws.data_ptr(), WS_BYTES, 0
    check(lib.ph_layernorm_bwd(C.byref(a), _stream()), 'ph_layernorm_bwd')
    return dx, (dx_drop if dx_drop is not None else dx)


def _attn_args(q, k, v, o, B, H, Sq, Sk, dh, strides, scale, key_mask, causal, drop, lse):
    f = _lib.AttnFwdArgs()
    f.q, f.k, f.v, f.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    (f.q_bs, f.q_ts), (f.k_bs, f.k_ts), (f.v_bs, f.v_ts), (f.o_bs, f.o_ts) = strides
    f.B, f.H, f.Sq, f.Sk, f.dh = B, H, Sq, Sk, dh
    f.scale = scale
    f.key_mask = ptr(key_mask)
    f.causal = int(causal)
    if drop is not None and drop.p > 0.0:
        f.drop_p, f.drop_seed, f.drop_stream = drop.p, drop.seed.data_ptr(), drop.stream
    else:
        f.drop_p, f.drop_seed, f.drop_stream = 0.0, None, 0
    f.lse = ptr(lse)
    return f


def attention_fwd(q, k, v, B, H, Sq, Sk, dh, *, q_strides, k_strides, v_strides, key_mask=None, causal=False, drop=None,
                  out=None):
    """q/k/v: any tensors whose (batch, token) strides (in elements) are given; heads contiguous (h*dh + c).
    Returns (o [B*Sq, H*dh] bf16, lse [B*H*Sq] fp32)."""
    if out is None:
        out = torch.empty((B * Sq, H * dh), dtype=BF16, device=q.device)
    lse = torch.empty((B * H * Sq,), dtype=F32, device=q.device)
    o_strides = (Sq * H * dh, H * dh)
    f = _attn_args(q, k, v, out, B, H, Sq, Sk, dh, (q_strides, k_strides, v_strides, o_strides), dh ** -0.5, key_mask,
                   causal, drop, lse)
    check(lib.ph_attention_fwd(C.byref(f), _stream()), 'ph_attention_fwd')
    return out, lse


def attention_bwd(d_o, q, k, v, o, lse, B, H, Sq, Sk, dh, *, q_strides, k_strides, v_strides, dq, dk, dv, dq_strides,
                  dk_strides, dv_strides, key_mask=None, causal=False, drop=None):
    a = _lib.AttnBwdArgs()
    o_strides = (Sq * H * dh, H * dh)
    a.f = _attn_args(q, k, v, o, B, H, Sq, Sk, dh, (q_strides, k_strides, v_strides, o_strides), dh ** -0.5, key_mask,
                     causal, drop, lse)
    a.d_o = d_o.data_ptr()
    a.do_bs, a.do_ts = o_strides
    a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    (a.dq_bs, a.dq_ts), (a.dk_bs, a.dk_ts), (a.dv_bs, a.dv_ts) = dq_strides, dk_strides, dv_strides
    delta = torch.empty((B * H * Sq,), dtype=F32, device=q.device)
    a.delta = delta.data_ptr()
    check(lib.ph_attention_bwd(C.byref(a), _stream()), 'ph_attention_bwd')


# ---------------------------------------------------------------------------------------------- front end

def patchify(img, p, Kp):
    B, Cc, R, _ = img.shape
    g = R // p
    col = torch.empty((B * g * g, Kp), dtype=BF16, device=img.device)
    check(lib.ph_patchify(img.data_ptr(), col.data_ptr(), B, Cc, R, p, Kp, _stream()), 'ph_patchify')
    return col


def resize_to_nhwc(x, Hout, Wout):
    B, Cc, Hin, Win = x.shape
    y = torch.empty((B, Hout, Wout, Cc), dtype=BF16, device=x.device)
    check(lib.ph_resize_bilinear_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), B, Cc, Hin, Win, Hout, Wout, _stream()),
          'ph_resize_bilinear_nchw_to_nhwc')
    return y


REMAP_PARTS = 16


def remap_resize_to_nhwc(raw, Hout, Wout):
    """raw: fp32 [B, C, H, W] dense expert map BEFORE post_label_process (depth / normal / edge).  Returns the stem input [B, Hout, Wout, C] bf16 =
    bilinear(2 * (x - min) / (max - min + 1e-6) - 1) with the min / max of each sample's whole map (dataset/utils.py:120-121 + vit.py:88-90)."""
    B, Cc, Hin, Win = raw.shape
    raw = raw.contiguous().float()
    part = torch.empty((B, REMAP_PARTS, 2), dtype=F32, device=raw.device)
    check(lib.ph_dense_minmax_partial(raw.data_ptr(), part.data_ptr(), B, Cc * Hin * Win, REMAP_PARTS, _stream()), 'ph_dense_minmax_partial')
    y = torch.empty((B, Hout, Wout, Cc), dtype=BF16, device=raw.device)
    check(lib.ph_resize_remap_nchw_to_nhwc(raw.data_ptr(), part.data_ptr(), REMAP_PARTS, y.data_ptr(), B, Cc, Hin, Win, Hout, Wout, _stream()),
          'ph_resize_remap_nchw_to_nhwc')
    return y


def inpaint_resize(label_map, table, Hout, Wout):
    """label_map: uint8 [B, H, W] (or [B, 1, H, W]); table: fp32 [256, C] (shared) or [B, 256, C] (per image).
    Returns the stem input [B, Hout, Wout, C] bf16 = bilinear(post_label_process(label_map)) (dataset/utils.py:117-160 + vit.py:88-90)."""
    if label_map.dim() == 4:
        label_map = label_map[:, 0]
    assert label_map.dtype == torch.uint8 and table.dtype == F32 and table.shape[-2] == 256
    label_map, table = label_map.contiguous(), table.contiguous()
    B, Hin, Win = label_map.shape
    Cc = table.shape[-1]
    stride = 256 * Cc if table.dim() == 3 else 0
    assert table.dim() == 2 or table.shape[0] == B
    y = torch.empty((B, Hout, Wout, Cc), dtype=BF16, device=label_map.device)
    check(lib.ph_inpaint_resize_nhwc(label_map.data_ptr(), table.data_ptr(), stride, y.data_ptr(), B, Cc, Hin, Win, Hout, Wout, _stream()),
          'ph_inpaint_resize_nhwc')
    return y


def conv_gemm_args(x, geo, w_shadow, y, col_stats=None):
    """GemmArgs of one implicit-GEMM forward convolution: y[B*Ho*Wo, Cout] = im2col(x) . w_shadow[Cout, Kp]^T.
    x: NHWC bf16 activation, geo = (B, H, W, C, ks, stride).  Returns (args, keep-alive tuple)."""
    B, H, W, Cc, ks, stride = geo
    Ho, Wo = conv_out_size(H, ks, stride), conv_out_size(W, ks, stride)
    g = _lib.GemmArgs()
    cg = _lib.ConvGather(B, H, W, Cc, ks, stride)
    g.A, g.B, g.C = x.data_ptr(), w_shadow.data_ptr(), y.data_ptr()
    g.M, g.N, g.K = B * Ho * Wo, w_shadow.shape[0], w_shadow.shape[1]
    g.lda, g.ldb, g.ldc = w_shadow.shape[1], w_shadow.stride(0), y.stride(0)
    g.alpha = 1.0
    g.conv = C.pointer(cg)
    g.col_stats = ptr(col_stats)
    return g, (cg, x, w_shadow, y, col_stats)


def conv_fwd_grouped(items):
    """items: [(x, geo, w_shadow, y, col_stats)] -- the same-layer convolutions of several expert stems in ONE launch per
    PH_GEMM_GROUP_MAX problems (vit.py:88-120: the stems are independent until their tokens are concatenated)."""
    for i0 in range(0, len(items), _lib.GEMM_GROUP_MAX):
        part = items[i0:i0 + _lib.GEMM_GROUP_MAX]
        arr = (_lib.GemmArgs * len(part))()
        keep = []
        for i, it in enumerate(part):
            g, k = conv_gemm_args(*it)
            arr[i] = g
            keep.append(k)
        check(lib.ph_gemm_grouped_bf16(arr, len(part), _stream()), 'ph_gemm_grouped_bf16 (conv)')


def wgrad_split_grouped(items):
    """Weight gradients with few output tiles and very long reductions -- the same conv layer of several expert stems -- in ONE grouped
    launch that is also split over K, plus one grouped fold pass (include/prismer_hip.h: ph_gemm_grouped_bf16 with a workspace).
    items: [(dy [K, M] bf16, x, M, N, K, conv)]: out[M, N] (fp32, returned) = dy^T . x with x = [K, N] bf16 (conv None) or the im2col
    view of the NHWC activation x (conv = (B, H, W, C, ks, stride): the gathered operand sits on the reduction side).
    Implicit and plain problems cannot share a launch, so they are grouped separately."""
    outs = [None] * len(items)
    for want_conv in (True, False):
        idx = [i for i, it in enumerate(items) if (it[5] is not None) == want_conv]
        for i0 in range(0, len(idx), _lib.GEMM_GROUP_MAX):
            part = idx[i0:i0 + _lib.GEMM_GROUP_MAX]
            arr = (_lib.GemmArgs * len(part))()
            keep = []
            ws = _workspace(items[part[0]][0].device, 1)
            for g, i in zip(arr, part):
                dy, x, M, N, K, conv = items[i]
                out = torch.empty((M, N), dtype=F32, device=dy.device)
                outs[i] = out
                g.A, g.B, g.C = dy.data_ptr(), x.data_ptr(), out.data_ptr()
                g.M, g.N, g.K = M, N, K
                g.lda, g.ldb, g.ldc = dy.stride(0), (N if conv is not None else x.stride(0)), out.stride(0)
                g.trans_a, g.trans_b, g.out_f32, g.alpha = 1, 1, 1, 1.0
                if conv is not None:
                    cg = _lib.ConvGather(*conv)
                    g.conv = C.pointer(cg)
                    keep.append(cg)
                g.workspace, g.workspace_bytes = ws.data_ptr(), WS_BYTES
            check(lib.ph_gemm_grouped_bf16(arr, len(part), _stream()), 'ph_gemm_grouped_bf16 (split wgrad)')
    return outs


def conv_out_size(H, ks, stride):
    pad = ks // 2
    return (H + 2 * pad - ks) // stride + 1


def im2col(x, B, H, W, Cc, ks, stride, Kp, bn_scale=None, bn_shift=None):
    Ho, Wo = conv_out_size(H, ks, stride), conv_out_size(W, ks, stride)
    col = torch.empty((B * Ho * Wo, Kp), dtype=BF16, device=x.device)
    check(lib.ph_im2col_nhwc(x.data_ptr(), col.data_ptr(), B, H, W, Cc, ks, stride, Kp, ptr(bn_scale), ptr(bn_shift), _stream()),
          'ph_im2col_nhwc')
    return col


def col2im(dcol, B, H, W, Cc, ks, stride, Kp):
    dx = torch.empty((B * H * W, Cc), dtype=BF16, device=dcol.device)
    check(lib.ph_col2im_nhwc(dcol.data_ptr(), dx.data_ptr(), B, H, W, Cc, ks, stride, Kp, _stream()), 'ph_col2im_nhwc')
    return dx


def bn_stats(y, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, out=None):
    """returns stats [4, C] fp32 = (mean, rstd, scale, shift); y: bf16 [M, C] conv output.
    out: optional ZEROED [4, C] slice of a per-step arena (saves the per-layer memset launch)."""
    M, Cc = y.shape
    st = out if out is not None else torch.empty((4, Cc), dtype=F32, device=y.device)
    check(lib.ph_bn_stats(y.data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                          momentum, eps, int(training), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                          int(out is not None), _stream()), 'ph_bn_stats')
    return st


def bn_relu_bwd(da, y, gamma, beta, stats, dgamma, dbeta, sums=None):
    """sums: optional ZEROED fp32 [2C] slice of a per-step arena (saves the per-layer memset launch)."""
    M, Cc = y.shape
    dy = torch.empty_like(y)
    pre = sums is not None
    if sums is None:
        sums = torch.empty((2 * Cc,), dtype=F32, device=y.device)
    check(lib.ph_bn_relu_bwd(da.data_ptr(), y.data_ptr(), dy.data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(), stats[0].data_ptr(),
                             stats[1].data_ptr(), ptr(dgamma), ptr(dbeta), sums.data_ptr(), int(pre), _stream()), 'ph_bn_relu_bwd')
    return dy


def _bn_items(items):
    out = []
    for i0 in range(0, len(items), _lib.BN_GROUP_MAX):
        part = items[i0:i0 + _lib.BN_GROUP_MAX]
        arr = (_lib.BnItem * len(part))()
        for it, d in zip(arr, part):
            it.y, it.a, it.dy = d['y'].data_ptr(), d['a'].data_ptr(), ptr(d.get('dy'))
            it.M, it.C = d['y'].shape[0], d['y'].shape[1]
            it.gamma, it.beta = d['gamma'].data_ptr(), d['beta'].data_ptr()
            it.running_mean, it.running_var = ptr(d.get('running_mean')), ptr(d.get('running_var'))
            it.stats, it.sums = d['stats'].data_ptr(), d['sums'].data_ptr()
            it.dgamma, it.dbeta = ptr(d.get('dgamma')), ptr(d.get('dbeta'))
        out.append((arr, len(part)))
    return out


def bn_apply_relu_grouped(items, training, momentum=0.1, eps=1e-5):
    """items: dicts y [M,C] bf16 (conv output), a [M,C] bf16 (out: relu(bn(y))), gamma, beta, running_mean, running_var,
    stats [4,C] fp32 (out), sums [PH_COLSTAT_SLABS = 8][2][C] fp64 (column sums / sums of squares from the conv GEMM epilogue,
    ph_gemm_args.col_stats; unused in eval mode)"""
    for arr, n in _bn_items(items):
        check(lib.ph_bn_apply_relu_grouped(arr, n, momentum, eps, int(training), _stream()), 'ph_bn_apply_relu_grouped')


def bn_relu_bwd_grouped(items):
    """items: dicts y, a (= dA, gradient w.r.t. relu(bn(y))), dy (out), gamma, beta, stats [4,C], sums [2,C] ZEROED, dgamma, dbeta"""
    for arr, n in _bn_items(items):
        check(lib.ph_bn_relu_bwd_grouped(arr, n, _stream()), 'ph_bn_relu_bwd_grouped')


def gemm_grouped(problems, trans_a=False, trans_b=False):
    """problems: [(a, b, out, M, N, K)] plain bf16 GEMMs of one layout in grouped launches (out bf16, no epilogue)"""
    for i0 in range(0, len(problems), _lib.GEMM_GROUP_MAX):
        part = problems[i0:i0 + _lib.GEMM_GROUP_MAX]
        arr = (_lib.GemmArgs * len(part))()
        for g, (a, b, out, M, N, K) in zip(arr, part):
            g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
            g.M, g.N, g.K = M, N, K
            g.lda, g.ldb, g.ldc = a.stride(0), b.stride(0), out.stride(0)
            g.trans_a, g.trans_b, g.alpha = int(trans_a), int(trans_b), 1.0
        check(lib.ph_gemm_grouped_bf16(arr, len(part), _stream()), 'ph_gemm_grouped_bf16')


def tokens_finalize(feat, pos, tokens, B, G, D, tok_per_batch, tok_off, inst=None, E=0, g=0, table=None, inst_emb=None):
    check(lib.ph_tokens_finalize(feat.data_ptr(), pos.data_ptr(), tokens.data_ptr(), B, G, D, tok_per_batch, tok_off, ptr(inst), E, g,
                                 ptr(table), ptr(inst_emb), _stream()), 'ph_tokens_finalize')


def tokens_finalize_bwd(dtokens, dfeat, dpos, B, G, D, tok_per_batch, tok_off, inst=None, E=0, g=0, table=None, dinst_emb=None):
    check(lib.ph_tokens_finalize_bwd(dtokens.data_ptr(), ptr(dfeat), ptr(dpos), B, G, D, tok_per_batch, tok_off, ptr(inst), E, g,
                                     ptr(table), ptr(dinst_emb), _stream()), 'ph_tokens_finalize_bwd')


def gather_taps(inp, idx, w, n_out, taps, D):
    out = torch.empty((n_out, D), dtype=F32, device=inp.device)
    check(lib.ph_gather_taps(inp.data_ptr(), out.data_ptr(), idx.data_ptr(), w.data_ptr(), n_out, taps, D, _stream()), 'ph_gather_taps')
    return out


def scatter_taps(dout, din, idx, w, n_out, taps, D):
    check(lib.ph_scatter_taps(dout.data_ptr(), din.data_ptr(), idx.data_ptr(), w.data_ptr(), n_out, taps, D, _stream()), 'ph_scatter_taps')


# ---------------------------------------------------------------------------------------------- decoder ends

def embed_args(ids, word, pos, typ, gamma, beta, eps, pad_id, out, xhat, rstd, drop, out_f32=None):
    B, T = ids.shape
    a = _lib.EmbedFwdArgs()
    a.ids, a.B, a.T, a.H, a.pad_id = ids.data_ptr(), B, T, word.shape[1], pad_id
    a.word, a.pos, a.type = word.data_ptr(), pos.data_ptr(), typ.data_ptr()
    a.gamma, a.beta, a.eps = gamma.data_ptr(), beta.data_ptr(), eps
    a.out, a.xhat, a.rstd = out.data_ptr(), ptr(xhat), ptr(rstd)
    if drop is not None and drop.p > 0.0:
        a.drop_p, a.drop_seed, a.drop_stream = drop.p, drop.seed.data_ptr(), drop.stream
    else:
        a.drop_p, a.drop_seed, a.drop_stream = 0.0, None, 0
    a.out_f32 = ptr(out_f32)
    return a


def embed_fwd(ids, word, pos, typ, gamma, beta, eps, pad_id, drop=None, want_f32=False):
    B, T = ids.shape
    H = word.shape[1]
    out = torch.empty((B * T, H), dtype=BF16, device=ids.device)
    xhat = torch.empty((B * T, H), dtype=BF16, device=ids.device)
    rstd = torch.empty((B * T,), dtype=F32, device=ids.device)
    out_f32 = torch.empty((B * T, H), dtype=F32, device=ids.device) if want_f32 else None
    a = embed_args(ids, word, pos, typ, gamma, beta, eps, pad_id, out, xhat, rstd, drop, out_f32)
    check(lib.ph_embed_fwd(C.byref(a), _stream()), 'ph_embed_fwd')
    if want_f32:
        return out, xhat, rstd, out_f32
    return out, xhat, rstd


def embed_bwd(dout, ids, word, pos, typ, gamma, beta, eps, pad_id, xhat, rstd, drop, dword, dpos, dtype, dgamma, dbeta):
    a = _lib.EmbedBwdArgs()
    a.f = embed_args(ids, word, pos, typ, gamma, beta, eps, pad_id, dout, xhat, rstd, drop)
    a.dout = dout.data_ptr()
    a.dword, a.dpos, a.dtype, a.dgamma, a.dbeta = ptr(dword), ptr(dpos), ptr(dtype), ptr(dgamma), ptr(dbeta)
    check(lib.ph_embed_bwd(C.byref(a), _stream()), 'ph_embed_bwd')


def ce_fwd(logits, labels, B, T, V, eps):
    loss = torch.empty((B,), dtype=F32, device=logits.device)
    row_lse = torch.empty((2, B * T), dtype=F32, device=logits.device)           # [0]: log-sum-exp per token (kept for the backward), [1]: per-token loss scratch
    check(lib.ph_ce_fwd(logits.data_ptr(), logits.stride(0), labels.data_ptr(), B, T, V, eps, loss.data_ptr(), row_lse[0].data_ptr(),
                        row_lse[1].data_ptr(), _stream()), 'ph_ce_fwd')
    row_lse = row_lse[0]
    return loss, row_lse


def softmax_gather(logits, ids):
    """softmax(logits, dim=1).index_select(1, ids) for bf16 logit rows [R, V] (a strided view of the decoder's padded logits buffer is fine:
    the row stride must be a multiple of 8 elements, the rows 16-B aligned); fp32 [R, len(ids)] out -- prismer_caption.py:70 / prismer_vqa.py:51"""
    assert logits.dim() == 2 and logits.dtype == BF16 and logits.stride(1) == 1 and ids.dtype == torch.int64
    R, V = logits.shape
    ids = ids.contiguous()
    out = torch.empty((R, ids.numel()), dtype=F32, device=logits.device)
    check(lib.ph_softmax_gather_bf16(logits.data_ptr(), logits.stride(0), R, V, ids.data_ptr(), ids.numel(), out.data_ptr(), _stream()), 'ph_softmax_gather_bf16')
    return out


def ce_bwd(logits, labels, B, T, V, eps, row_lse, dloss):
    check(lib.ph_ce_bwd(logits.data_ptr(), logits.stride(0), labels.data_ptr(), B, T, V, logits.shape[1], eps, row_lse.data_ptr(),
                        dloss.data_ptr(), _stream()), 'ph_ce_bwd')
    return logits


# ---------------------------------------------------------------------------------------------- utilities

def adamw(p, g, m, v, p_bf16, n, hyper, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.05, grad_scale=1.0, zero_grad=False, keep=None):
    """keep: optional int32 tensor, bit c = gradients [1024 c, 1024 c + 1024) keep their values (see ph_adamw_keep)"""
    assert keep is None or keep.numel() * 32 * 1024 >= n
    check(lib.ph_adamw_keep(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), ptr(p_bf16), n, hyper.data_ptr(), beta1, beta2, eps,
                            weight_decay, grad_scale, int(zero_grad), ptr(keep), _stream()), 'ph_adamw')


def cast_to_bf16(x, out=None, scale=1.0):
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    if scale == 1.0:
        check(lib.ph_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), 'ph_cast_f32_to_bf16')
    else:
        check(lib.ph_scale_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), float(scale), _stream()), 'ph_scale_cast_f32_to_bf16')
    return out


def cast_to_f32(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=F32, device=x.device)
    check(lib.ph_cast_bf16_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), 'ph_cast_bf16_to_f32')
    return out


def colsum(x, out, M=None, N=None):
    """out[N] (fp32) += column sums of bf16 x[M, N]."""
    M = x.shape[0] if M is None else M
    N = x.shape[1] if N is None else N
    check(lib.ph_colsum_bf16(x.data_ptr(), M, N, x.stride(0), out.data_ptr(), _stream()), 'ph_colsum_bf16')


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(lib.ph_add_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), 'ph_add_bf16')
    return out


def act_bwd(dy, pre, act, out=None):
    if out is None:
        out = torch.empty_like(dy)
    check(lib.ph_act_bwd_bf16(dy.data_ptr(), pre.data_ptr(), out.data_ptr(), dy.numel(), act, _stream()), 'ph_act_bwd_bf16')
    return out


def copy_rows(src, dst, rows, cols, src_map=IDENT, dst_map=IDENT, accumulate=False, src_ld=None, dst_ld=None):
    check(lib.ph_copy_rows_bf16(src.data_ptr(), src_ld or src.stride(0), src_map, dst.data_ptr(), dst_ld or dst.stride(0), dst_map,
                                rows, cols, int(accumulate), _stream()), 'ph_copy_rows_bf16')


def gather_rows(src, idx, dst, cols=None):
    """dst[r, :cols] = src[idx[r], :cols]; src / dst: 2-D bf16 views (row stride = stride(0)), idx: int32 device tensor"""
    assert src.dtype == BF16 and dst.dtype == BF16 and idx.dtype == torch.int32 and idx.is_cuda
    cols = src.shape[1] if cols is None else cols
    check(lib.ph_gather_rows_bf16(src.data_ptr(), src.stride(0), idx.data_ptr(), dst.data_ptr(), dst.stride(0), idx.numel(), cols, _stream()),
          'ph_gather_rows_bf16')
    return dst


def conv_weight_to_shadow(w, shadow, Cout, Cin, ks, Kp):
    check(lib.ph_conv_weight_to_shadow(w.data_ptr(), shadow.data_ptr(), Cout, Cin, ks, Kp, _stream()), 'ph_conv_weight_to_shadow')


def conv_grad_from_shadow(dshadow, dw, Cout, Cin, ks, Kp):
    check(lib.ph_conv_grad_from_shadow(dshadow.data_ptr(), dw.data_ptr(), Cout, Cin, ks, Kp, _stream()), 'ph_conv_grad_from_shadow')


def conv_layout_grouped(items, to_shadow):
    """items: [(src, dst, Cout, Cin, ks, Kp)]; one launch per PH_CONV_GROUP_MAX layers."""
    fn = lib.ph_conv_weight_to_shadow_grouped if to_shadow else lib.ph_conv_grad_from_shadow_grouped
    for i0 in range(0, len(items), _lib.CONV_GROUP_MAX):
        part = items[i0:i0 + _lib.CONV_GROUP_MAX]
        arr = (_lib.ConvLayoutItem * len(part))()
        for it, (src, dst, Co, Ci, ks, Kp) in zip(arr, part):
            it.src, it.dst, it.Cout, it.Cin, it.ks, it.Kp = src.data_ptr(), dst.data_ptr(), Co, Ci, ks, Kp
        check(fn(arr, len(part), _stream()), 'ph_conv_layout_grouped')


def conv_dgrad_shadows(items):
    """items: [(w fp32 [Cout,Cin,3,3], dst bf16 [Cin, 9*Cout], Cout, Cin, stride)]: weight operands of the implicit data gradients,
    one launch per PH_CONV_GROUP_MAX layers (include/prismer_hip.h: ph_conv_dgrad_shadow_grouped)"""
    for i0 in range(0, len(items), _lib.CONV_GROUP_MAX):
        part = items[i0:i0 + _lib.CONV_GROUP_MAX]
        arr = (_lib.ConvDgradItem * len(part))()
        for it, (w, dst, Co, Ci, stride) in zip(arr, part):
            it.w, it.dst, it.Cout, it.Cin, it.stride = w.data_ptr(), dst.data_ptr(), Co, Ci, stride
        check(lib.ph_conv_dgrad_shadow_grouped(arr, len(part), _stream()), 'ph_conv_dgrad_shadow_grouped')


# parity classes of the input pixel of a stride-2, 3x3, pad-1 convolution: (py, px) -> (first tap, taps) in the dgrad shadow
DGRAD_CLASSES = (((0, 0), 0, 1), ((0, 1), 1, 2), ((1, 0), 3, 2), ((1, 1), 5, 4))


def conv_dgrad_grouped(items):
    """Implicit data gradients of 3x3 (pad 1) convolutions, the same layer of several stems in grouped launches (round 3; replaces
    dcol = dY . W followed by the col2im gather: 9 x the output in transient traffic and, for stride 2, 4 x the useful FLOPs).
    items: [(dy [B*Ho*Wo, Cout] bf16, wd [Cin, 9*Cout] dgrad shadow, dx [B*H*W, Cin] bf16 out, (B, H, W, Cin, Cout, stride))].
    stride 1: dX = conv3x3(dY) with flipped taps, one problem.  stride 2: one problem per parity class of the input pixel: the
    A operand gathers a (1+py) x (1+px) window of dY, the output rows are scattered to the class's pixels (GemmArgs.rowmap_*)."""
    probs, keep = [], []
    for dy, wd, dx, (B, H, W, Ci, Co, stride) in items:
        Ho, Wo = (H, W) if stride == 1 else (H // 2, W // 2)
        if stride == 1:
            classes = (((0, 0), 0, 9),)
        else:
            assert H % 2 == 0 and W % 2 == 0 and stride == 2
            classes = DGRAD_CLASSES
        for (py, px), t0, nt in classes:
            g = _lib.GemmArgs()
            if stride == 1:
                cg = _lib.ConvGather(B, Ho, Wo, Co, 3, 1, 3, 3, -1, -1, Ho, Wo)
            else:
                cg = _lib.ConvGather(B, Ho, Wo, Co, 3, 1, 1 + py, 1 + px, 0, 0, Ho, Wo)
                g.rowmap_wo, g.rowmap_mul, g.rowmap_sub, g.rowmap_add = Wo, 4, 2, py * W + px
            g.A, g.B, g.C = dy.data_ptr(), wd.data_ptr() + 2 * t0 * Co, dx.data_ptr()
            g.M, g.N, g.K = B * Ho * Wo, Ci, nt * Co
            g.lda, g.ldb, g.ldc = nt * Co, wd.stride(0), dx.stride(0)
            g.alpha = 1.0
            g.conv = C.pointer(cg)
            probs.append(g)
            keep.append((cg, dy, wd, dx))
    for i0 in range(0, len(probs), _lib.GEMM_GROUP_MAX):
        part = probs[i0:i0 + _lib.GEMM_GROUP_MAX]
        arr = (_lib.GemmArgs * len(part))()
        for i, g in enumerate(part):
            arr[i] = g
        check(lib.ph_gemm_grouped_bf16(arr, len(part), _stream()), 'ph_gemm_grouped_bf16 (conv dgrad)')


def add_i64(x, value=1):
    """x (int64, contiguous) += value -- BatchNorm's num_batches_tracked counters, all layers in one launch"""
    check(lib.ph_add_i64(x.data_ptr(), x.numel(), int(value), _stream()), 'ph_add_i64')


def weighted_sum(x, weights, scale, out=None):
    """out[0] = scale * sum(x * weights) (weights None: plain sum): the batch loss"""
    if out is None:
        out = torch.empty(1, dtype=F32, device=x.device)
    check(lib.ph_weighted_sum_f32(x.data_ptr(), ptr(weights), x.numel(), float(scale), out.data_ptr(), _stream()), 'ph_weighted_sum_f32')
    return out


_ELEM_SIZE = {}


def zeros(shape, dtype, device):
    """torch.zeros through the library: an empty tensor + ONE ph_fill_zero launch (no stock torch kernel, no memset node in a captured step)."""
    if isinstance(shape, int):
        shape = (shape,)
    n = 1
    for v in shape:
        n *= int(v)
    esz = _ELEM_SIZE.get(dtype)
    if esz is None:
        esz = _ELEM_SIZE[dtype] = (torch.finfo(dtype).bits if dtype.is_floating_point else (8 if dtype == torch.bool else torch.iinfo(dtype).bits)) // 8
    nbytes = n * esz
    dev = device if isinstance(device, torch.device) else torch.device(device)
    # the fill runs on the calling thread's current stream: that stream must belong to `device` (round-5 advisor finding)
    assert dev.type != 'cuda' or dev.index is None or dev.index == torch.cuda.current_device(), 'ops.zeros: `device` is not the current device'
    pad = (nbytes + 15) // 16 * 16
    buf = torch.empty(pad, dtype=torch.uint8, device=device)              # (the caching allocator aligns to 512 B)
    check(lib.ph_fill_zero(buf.data_ptr(), pad, _stream()), 'ph_fill_zero')
    return buf[:nbytes].view(dtype).reshape(shape)


def copy_flat(dst, src):
    """dst[:] = src for contiguous 1-D tensors of one dtype: ph_copy_bytes for the 16-B aligned body, Tensor.copy_ for a ragged tail / head"""
    assert dst.dtype == src.dtype and dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
    nbytes = dst.numel() * dst.element_size()
    if nbytes == 0:
        return dst
    if (dst.data_ptr() | src.data_ptr()) & 15 or nbytes < 16:
        dst.copy_(src)
        return dst
    body = nbytes // 16 * 16
    check(lib.ph_copy_bytes(dst.data_ptr(), src.data_ptr(), body, _stream()), 'ph_copy_bytes')
    if body < nbytes:
        k = body // dst.element_size()
        dst.reshape(-1)[k:].copy_(src.reshape(-1)[k:])
    return dst


STORE_WORDS_MAX = 320


def store_words(dst0, vals0, dst1=None, vals1=()):
    """dst0[:len(vals0)] (fp32 tensor) <- floats, dst1[:len(vals1)] (int32 tensor) <- ints, carried in the arguments of one tiny kernel
    (no host-to-device copy: see ph_store_words)"""
    n0, n1 = len(vals0), len(vals1)
    assert dst0.dtype == F32 and dst0.numel() >= n0 and (n1 == 0 or (dst1.dtype == torch.int32 and dst1.numel() >= n1))
    words = struct.pack(f'{n0}f{n1}i', *vals0, *vals1)
    buf = (C.c_uint32 * (n0 + n1)).from_buffer_copy(words)
    check(lib.ph_store_words(dst0.data_ptr(), n0, ptr(dst1) if n1 else None, n1, buf, _stream()), 'ph_store_words')


def advance_seed(seed):
    check(lib.ph_advance_seed(seed.data_ptr(), _stream()), 'ph_advance_seed')


def probe_layouts(inp):
    out = torch.empty((512 + 1024,), dtype=F32, device=inp.device)
    check(lib.ph_probe_layouts(inp.data_ptr(), out.data_ptr(), _stream()), 'ph_probe_layouts')
    return out
