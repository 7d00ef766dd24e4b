"""Layer programs (forward + hand-written backward) of the Prismer vision side on the HIP operator set.

What the reference expresses as an nn.Module graph walked by autograd
  VisionTransformer.forward        model/modules/vit.py:133-172
  expert stems                     vit.py:88-120
  PerceiverResampler               model/modules/resampler.py:33-52
  Transformer / Adaptor            vit.py:70-75, model/modules/utils.py:48-65
is here a straight-line program of C-ABI kernel launches; the backward program replays the saved activations in
reverse and writes weight gradients directly into the flat fp32 gradient buffer (prismer_amd/store.py).
Internal activation layout is batch-major [B*tokens, D] bf16; the [S, B, D] view the reference returns is a
permuted view of the final buffer.
"""
import torch
import torch.nn.functional as F

from .. import ops
from .._lib import ACT_QUICKGELU, ACT_RELU2, ACT_SAVED_GRAD, COLSTAT_SLABS, IDENT, RowMap
from ..config import LABEL_DOMAINS

BF16, F32 = torch.bfloat16, torch.float32


# Activations inside a Linear: the forward epilogue stores act'(x) (bf16, taken from the fp32 accumulator) instead of x, and the
# backward GEMM multiplies by it -- nothing in the backward needs x itself, and its epilogue loses two transcendentals per
# element (QuickGELU' / erf-GELU').
SAVE_ACT_GRAD = True


def _bwd_act(act):
    return ACT_SAVED_GRAD


def _rup(x, m):
    return (x + m - 1) // m * m


class Linear:
    """y = act(x W^T + b) bookkeeping shared by all programs: forward GEMM, dgrad GEMM, wgrad GEMM + bias colsum."""

    def __init__(self, P, wname, bname=None, rows=None, cols=None):
        self.P, self.wname, self.bname = P, wname, bname
        shp = P.shape[wname]
        self.rows = rows or shp[0]
        self.cols = cols or shp[1]

    @property
    def w(self):
        return self.P.w2(self.wname, self.rows, self.cols)

    @property
    def b(self):
        return self.P.fvec(self.bname, self.rows) if self.bname else None

    def fwd(self, x, **kw):
        return ops.gemm(x, self.w, bias=self.b, **kw)

    def dgrad(self, dy, **kw):
        """dx = dy . W   (W stored [out][in] = K-strided operand)"""
        return ops.gemm(dy, self.w, trans_b=True, N=self.cols, K=self.rows, **kw)

    def wgrad(self, dy, x):
        """weight / bias gradients: deferred into the grouped-launch queue (ops.WgradQueue)"""
        gw = self.P.g2(self.wname, self.rows, self.cols)
        gb = self.P.gvec(self.bname, self.rows) if self.bname else None
        if gw is None and gb is None:
            return

        if gw is not None:
            ops.WQ.add_gemm(dy, x, gw, self.rows, self.cols, dy.shape[0])
        if gb is not None:
            ops.WQ.add_colsum(dy, gb, self.rows)


class LN:
    def __init__(self, P, prefix, eps=1e-5):
        self.P, self.wn, self.bn, self.eps = P, prefix + '.weight', prefix + '.bias', eps

    def fwd(self, x, **kw):
        return ops.layernorm_fwd(x, self.P.f(self.wn), self.P.f(self.bn), self.eps, **kw)

    def bwd(self, dy, x, mean, rstd, **kw):
        return ops.layernorm_bwd(dy, x, mean, rstd, self.P.f(self.wn), dgamma=self.P.g(self.wn), dbeta=self.P.g(self.bn), **kw)


class EncoderProgram:
    def __init__(self, module, dims, store):
        self.mod, self.d, self.P = module, dims, store
        d, P = dims, store
        # BatchNorm's num_batches_tracked of every stem layer live in ONE int64 buffer (the module buffers are 0-dim views of it, so
        # state_dict / load_state_dict see them unchanged): a training forward increments all of them with one ph_add_i64 launch
        # (only the stems of the experts present in a batch advance, like the reference's per-module counters: _bn_range[dom] = rows of dom)
        bns, self._bn_range = [], {}
        for dom in getattr(module, 'conv1', {}):
            if dom == 'rgb':
                continue
            mine = [m for m in module.conv1[dom] if isinstance(m, torch.nn.BatchNorm2d)]
            self._bn_range[dom] = (len(bns), len(mine))
            bns += mine
        self._bn_flat, self._bn_mods, self._bn_stats_flat = None, bns, None
        if bns:
            self._bn_flat = torch.stack([m.num_batches_tracked.reshape(()) for m in bns]).to(torch.int64).contiguous()
            for i, m in enumerate(bns):
                m._buffers['num_batches_tracked'] = self._bn_flat[i]
            # round 6: running_mean / running_var of every stem layer are views of ONE fp32 buffer as well (16-B aligned pieces), so that DDP's
            # per-step buffer broadcast (train_caption.py:117: DistributedDataParallel(broadcast_buffers=True)) is two collectives
            # -- this buffer and the counters -- instead of 72 (Trainer(broadcast_buffers=True))
            pieces, off = [], 0
            for m in bns:
                for nm in ('running_mean', 'running_var'):
                    t = m._buffers[nm]
                    pieces.append((m, nm, off, t.numel()))
                    off += (t.numel() + 3) // 4 * 4
            flat = torch.zeros(off, dtype=F32, device=self._bn_flat.device)
            for m, nm, o, n in pieces:
                flat[o:o + n].copy_(m._buffers[nm].reshape(-1).float())
                m._buffers[nm] = flat[o:o + n]
            self._bn_stats_flat = flat
        W = d.width
        self.Kp_rgb = _rup(3 * d.patch_size ** 2, 8)
        self.experts = [e for e in d.experts if e != 'rgb']
        self.blocks = []
        for l in range(d.vit_layers):
            p = f'transformer.resblocks.{l}.'
            self.blocks.append(dict(
                ln_1=LN(P, p + '0.ln_1'), ln_2=LN(P, p + '0.ln_2'), aln=LN(P, p + '1.adaptor_ln'),
                qkv=Linear(P, p + '0.attn.in_proj_weight', p + '0.attn.in_proj_bias'),
                out=Linear(P, p + '0.attn.out_proj.weight', p + '0.attn.out_proj.bias'),
                down=Linear(P, p + '1.adaptor.down_proj.weight', p + '1.adaptor.down_proj.bias'),
                up=Linear(P, p + '1.adaptor.up_proj.weight', p + '1.adaptor.up_proj.bias'),
                fc=Linear(P, p + '0.mlp.c_fc.weight', p + '0.mlp.c_fc.bias'),
                proj=Linear(P, p + '0.mlp.c_proj.weight', p + '0.mlp.c_proj.bias')))
        self.rblocks = []
        if d.has_experts:
            for l in range(d.resampler_layers):
                p = f'resampler.perceiver_blocks.{l}.'
                self.rblocks.append(dict(
                    ln_1=LN(P, p + 'ln_1'), ln_2=LN(P, p + 'ln_2'), ln_ff=LN(P, p + 'ln_ff'),
                    out=Linear(P, p + 'attn.out_proj.weight', p + 'attn.out_proj.bias'),
                    fc=Linear(P, p + 'mlp.c_fc.weight', p + 'mlp.c_fc.bias'),
                    proj=Linear(P, p + 'mlp.c_proj.weight', p + 'mlp.c_proj.bias'),
                    inw=p + 'attn.in_proj_weight', inb=p + 'attn.in_proj_bias'))
        self.ln_pre, self.ln_post = LN(P, 'ln_pre'), LN(P, 'ln_post')
        self._taps = None

    # ---------------------------------------------------------------------------------------- conv weight shadows
    def conv_shadow(self, name, ks):
        Co, Ci = self.P.shape[name][0], self.P.shape[name][1]
        Kp = _rup(ks * ks * Ci, 8)
        return self.P.derived_buffer(name, (Co, Kp), lambda t: ops.conv_weight_to_shadow(self.P.f(name), t, Co, Ci, ks, Kp),
                                     conv=(name, Co, Ci, ks, Kp)), Kp

    def conv_dgrad_shadow(self, name, stride):
        """[Cin, 9*Cout] weight operand of the implicit data gradient of a 3x3 conv (ops.conv_dgrad_grouped), re-derived with the
        other shadows after every optimizer step"""
        Co, Ci = self.P.shape[name][0], self.P.shape[name][1]
        return self.P.derived_buffer(name + '#dgrad', (Ci, 9 * Co), lambda t: ops.conv_dgrad_shadows([(self.P.f(name), t, Co, Ci, stride)]),
                                     conv=('dgrad', name, Co, Ci, stride))

    def conv_wgrad(self, name, ks, dy, col):
        """dW (shadow layout [Cout, Kp], fp32) = dy^T . col, folded back into the [Cout,Cin,k,k] gradient."""
        g = self.P.g(name)
        if g is None:
            return
        Co, Ci = self.P.shape[name][0], self.P.shape[name][1]
        Kp = col.shape[1]
        if ks == 1 and Kp == Ci:                         # 1x1 conv: the shadow layout IS the parameter layout
            ops.WQ.add_gemm(dy, col, g.view(Co, Ci), Co, Kp, dy.shape[0])
            return
        ds = ops.off_critical_path(lambda: ops.gemm(dy, col, trans_a=True, trans_b=True, out_f32=True, M=Co, N=Kp, K=dy.shape[0]), dy, col)
        ops.WQ.add_conv_fold(ds, g, Co, Ci, ks, Kp)      # folded back into [Cout, Cin, k, k] with all other layers at the flush

    # ---------------------------------------------------------------------------------------- positional embedding
    def expert_pos(self):
        """vit.py:157 / utils.py:34-44: bicubic (align_corners=False) re-grid when the expert grid differs from the rgb
        grid.  The interpolation is linear in the table: out = sum_t w[i,t] * pos[idx[i,t]] with <= 16 taps."""
        d = self.d
        pos = self.P.f('positional_embedding')
        if d.expert_grid == d.rgb_grid:
            return pos
        if self._taps is None:
            n, g = d.rgb_grid, d.expert_grid
            eye = torch.eye(n * n).reshape(1, n * n, n, n)           # basis images (host, once)
            Wm = F.interpolate(eye, size=(g, g), mode='bicubic', align_corners=False).reshape(n * n, g * g).t()
            w, idx = torch.topk(Wm.abs(), 16, dim=1)
            w = torch.gather(Wm, 1, idx)
            self._taps = (idx.to(torch.int32).contiguous().to(pos.device), w.contiguous().to(pos.device))
        idx, w = self._taps
        return ops.gather_taps(pos, idx, w, d.expert_grid ** 2, 16, d.width)

    def expert_pos_bwd(self, dpos_e):
        d = self.d
        g = self.P.g('positional_embedding')
        if g is None or d.expert_grid == d.rgb_grid:
            return
        idx, w = self._taps
        ops.scatter_taps(dpos_e, g, idx, w, d.expert_grid ** 2, 16, d.width)

    # ---------------------------------------------------------------------------------------- expert stems
    # ---------------------------------------------------------------------------------------- expert stems, grouped (round 2)
    # The six stems are independent networks of the same depth (vit.py:88-120).  They are walked LAYER by layer, all experts of a
    # layer in one grouped launch each:
    #   conv3x3 as an IMPLICIT GEMM (the A operand is gathered from the NHWC activation inside the GEMM loader: no im2col matrix is
    #   written, kept for the backward or re-read), BatchNorm statistics accumulated in the GEMM epilogue (col_stats), then one
    #   grouped pass a = relu(bn(y)) that also derives scale / shift and updates the running statistics.
    # Only the first layer of the dense stems (Cin = 1 or 3: K = 9 or 27) keeps an explicit (tiny) im2col matrix (direct kernels for it were
    # built in round 6 and measured slower: profiles/r6_ab_stem_conv1.txt).
    # Backward: grouped BN-ReLU backward; weight gradients as implicit GEMMs with the gather on the reduction side; data gradients
    # as dcol = dY.W followed by the col2im gather (dcol is transient, never saved).
    def stems_fwd(self, x, names, training, sv):
        d = self.d
        P = self.P
        st = []                                                   # per expert running state
        for name in names:
            dom = 'seg' if 'seg' in name else name
            val = x[name]
            label = dom in LABEL_DOMAINS
            Hs = int(d.expert_resolution * (4 if label else 16) / d.patch_size)
            if isinstance(val, dict) and 'label_map' in val:
                a = ops.inpaint_resize(val['label_map'], val['table'], Hs, Hs)
            elif isinstance(val, dict) and 'raw' in val:          # dense expert as the expert network left it: min-max remap on the device (round 6)
                a = ops.remap_resize_to_nhwc(val['raw'], Hs, Hs)
            else:
                inp = (val['label'] if name == 'obj_detection' else val).contiguous().float()
                a = ops.resize_to_nhwc(inp, Hs, Hs)
            st.append(dict(name=name, dom=dom, a=a, H=Hs, C=a.shape[3], B=a.shape[0], strides=(2, 2, 1, 1) if label else (2, 2, 2, 2),
                           a_in=[], ys=[], stats=[], geo=[], col0=None))
        n_ch = sum(d.width // k for k in (8, 4, 2, 1))
        SL = COLSTAT_SLABS
        sums_arena = ops.zeros(len(st) * SL * 2 * n_ch, torch.float64, st[0]['a'].device) if st else None     # fp64: see tile_colstats
        stats_arena = torch.empty(len(st) * 4 * n_ch, dtype=F32, device=st[0]['a'].device) if st else None
        so = to = 0
        for i in range(4):
            conv_items, bn_items = [], []
            for e in st:
                dom, B, H, C, s_ = e['dom'], e['B'], e['H'], e['C'], e['strides'][i]
                wname = f'conv1.{dom}.{1 + 3 * i}.weight'
                shadow, Kp = self.conv_shadow(wname, 3)
                Co = shadow.shape[0]
                Ho = ops.conv_out_size(H, 3, s_)
                y = torch.empty(B * Ho * Ho, Co, dtype=BF16, device=e['a'].device)
                sums = sums_arena[so:so + SL * 2 * Co].view(SL, 2, Co); so += SL * 2 * Co
                stats = stats_arena[to:to + 4 * Co].view(4, Co); to += 4 * Co
                if C % 8 == 0:
                    conv_items.append((e['a'], (B, H, H, C, 3, s_), shadow, y, sums if training else None))
                else:                                              # Cin = 1 / 3: explicit im2col (K = 16 / 32), statistics still fused
                    col = ops.im2col(e['a'], B, H, H, C, 3, s_, Kp)
                    ops.gemm(col, shadow, out=y, col_stats=sums if training else None)
                    e['col0'] = col
                bn = self.mod.conv1[dom][2 + 3 * i]
                a_next = torch.empty_like(y)
                bn_items.append(dict(y=y, a=a_next, gamma=P.f(f'conv1.{dom}.{2 + 3 * i}.weight'), beta=P.f(f'conv1.{dom}.{2 + 3 * i}.bias'),
                                     running_mean=bn.running_mean, running_var=bn.running_var, stats=stats, sums=sums))
                e['a_in'].append(e['a']); e['ys'].append(y); e['stats'].append(stats); e['geo'].append((H, C, s_, Kp))
                e['a'], e['H'], e['C'] = a_next.view(B, Ho, Ho, Co), Ho, Co
            if conv_items:
                ops.conv_fwd_grouped(conv_items)
            bn0 = self.mod.conv1[st[0]['dom']][2]
            ops.bn_apply_relu_grouped(bn_items, training, bn0.momentum, bn0.eps)
        feats, probs = [], []
        for e in st:
            shadow, Kp = self.conv_shadow(f'conv1.{e["dom"]}.13.weight', 1)
            M = e['B'] * e['H'] * e['H']
            a2 = e['a'].view(M, e['C'])
            f = torch.empty(M, shadow.shape[0], dtype=BF16, device=a2.device)
            probs.append((a2, shadow, f, M, shadow.shape[0], Kp))
            feats.append(f)
            if sv is not None:
                sv[e['dom']] = dict(a_in=e['a_in'], ys=e['ys'], stats=e['stats'], geo=e['geo'], a_last=a2, col0=e['col0'], B=e['B'], Hlast=e['H'])
        if probs:
            ops.gemm_grouped(probs)
        return feats

    def stems_bwd(self, doms, dfeats, sv):
        """doms: stem names in token order; dfeats: d(loss)/d(stem output) [B*G, W] per stem."""
        P = self.P
        dev = dfeats[0].device
        S = [sv[dom] for dom in doms]
        # 1x1 projection (slot 13): weight gradient deferred (plain TT GEMM), data gradient grouped
        das, probs = [], []
        for dom, s, df in zip(doms, S, dfeats):
            w13 = f'conv1.{dom}.13.weight'
            g = P.g(w13)
            Co, Ci = P.shape[w13][0], P.shape[w13][1]
            if g is not None:
                ops.WQ.add_gemm(df, s['a_last'], g.view(Co, Ci), Co, Ci, df.shape[0])
            shadow, _ = self.conv_shadow(w13, 1)
            da = torch.empty(df.shape[0], Ci, dtype=BF16, device=dev)
            probs.append((df, shadow, da, df.shape[0], Ci, Co))
            das.append(da)
        ops.gemm_grouped(probs, trans_b=True)
        n_ch = sum(self.d.width // k for k in (8, 4, 2, 1))
        sums_arena = ops.zeros(len(doms) * 2 * n_ch, F32, dev)
        so = 0
        for i in (3, 2, 1, 0):
            bn_items, dys = [], []
            for dom, s, da in zip(doms, S, das):
                y = s['ys'][i]
                Co = y.shape[1]
                gname, bname = f'conv1.{dom}.{2 + 3 * i}.weight', f'conv1.{dom}.{2 + 3 * i}.bias'
                dy = torch.empty_like(y)
                bn_items.append(dict(y=y, a=da, dy=dy, gamma=P.f(gname), beta=P.f(bname), stats=s['stats'][i],
                                     sums=sums_arena[so:so + 2 * Co], dgamma=P.g(gname), dbeta=P.g(bname)))
                so += 2 * Co
                dys.append(dy)
            ops.bn_relu_bwd_grouped(bn_items)
            dcol_probs, dcols = [], []
            for dom, s, dy in zip(doms, S, dys):
                H, C, stride, Kp = s['geo'][i]
                wname = f'conv1.{dom}.{1 + 3 * i}.weight'
                g = P.g(wname)
                Co = P.shape[wname][0]
                if g is not None:
                    if C % 8 == 0:                                 # implicit-GEMM weight gradient: the im2col view sits on the reduction side
                        ds = ops.gemm(dy, s['a_in'][i].view(-1, C), trans_a=True, trans_b=True, out_f32=True, M=Co, N=Kp, K=dy.shape[0],
                                      conv=(s['B'], H, H, C, 3, stride), defer_reduce=True)
                    else:
                        ds = ops.gemm(dy, s['col0'], trans_a=True, trans_b=True, out_f32=True, M=Co, N=Kp, K=dy.shape[0],
                                      defer_reduce=True)
                    ops.WQ.add_conv_fold(ds, g, Co, C, 3, Kp)
                if i > 0:
                    da = torch.empty(s['B'] * H * H, C, dtype=BF16, device=dev)
                    dcol_probs.append((dy, self.conv_dgrad_shadow(wname, stride), da, (s['B'], H, H, C, Co, stride)))
                    dcols.append(da)
            ops.gemm_flush_deferred()                            # the split-K fold passes of this layer's weight gradients: one grouped launch
            if i > 0:
                ops.conv_dgrad_grouped(dcol_probs)               # data gradients gathered from dY (no dcol matrix, no col2im pass)
                das = dcols

    # ---------------------------------------------------------------------------------------- resampler
    def resampler_fwd(self, xf, B, h, sv):
        """resampler.py:46-52; writes the 64 latents of every image into rows [b*S + N + l] of h."""
        d, P = self.d, self.P
        W, L, Mx = d.width, d.num_latents, d.num_expert_tokens
        KV = L + Mx
        H = d.resampler_heads
        dh = W // H
        lat = torch.empty(B * L, W, dtype=BF16, device=xf.device)
        ops.copy_rows(P.w('resampler.latents'), lat, B * L, W, src_map=RowMap(L, 0, 0))       # repeat 'l d -> l b d'
        layers = []
        for blk in self.rblocks:
            kvin = torch.empty(B * KV, W, dtype=BF16, device=xf.device)
            qin = torch.empty(B * L, W, dtype=BF16, device=xf.device)
            _, m1, r1 = blk['ln_1'].fwd(lat, out=kvin, out_map=RowMap(L, KV, 0), out2=qin)
            _, m2, r2 = blk['ln_2'].fwd(xf, out=kvin, out_map=RowMap(Mx, KV, L))
            q = ops.gemm(qin, P.w2(blk['inw'], W, W), bias=P.fvec(blk['inb'], W))
            kv = ops.gemm(kvin, P.w2(blk['inw'], 3 * W, W)[W:], bias=P.fvec(blk['inb'], 3 * W)[W:])
            ks = (KV * 2 * W, 2 * W)
            o, lse = ops.attention_fwd(q, kv[:, :W], kv[:, W:], B, H, L, KV, dh, q_strides=(L * W, W), k_strides=ks, v_strides=ks)
            lat1 = blk['out'].fwd(o, residual=lat)
            f, m3, r3 = blk['ln_ff'].fwd(lat1)
            hpre = torch.empty(B * L, 4 * W, dtype=BF16, device=xf.device)
            hact = blk['fc'].fwd(f, act=ACT_RELU2, pre_out=hpre, pre_grad=SAVE_ACT_GRAD)
            lat2 = blk['proj'].fwd(hact, residual=lat1)
            layers.append(dict(lat=lat, kvin=kvin, qin=qin, m1=m1, r1=r1, m2=m2, r2=r2, q=q, kv=kv, o=o, lse=lse, lat1=lat1, f=f,
                               m3=m3, r3=r3, hpre=hpre, hact=hact))
            lat = lat2
        ops.copy_rows(lat, h, B * L, W, dst_map=RowMap(L, d.seq_len, d.num_rgb_tokens))
        if sv is not None:
            sv['resampler'] = layers
            sv['xf'] = xf

    def resampler_bwd(self, dlat, B, sv, dxf_out=None):
        d, P = self.d, self.P
        W, L, Mx = d.width, d.num_latents, d.num_expert_tokens
        KV = L + Mx
        H = d.resampler_heads
        dh = W // H
        xf = sv['xf']
        dxf = None
        for blk, s in zip(reversed(self.rblocks), reversed(sv['resampler'])):
            dhpre = blk['proj'].dgrad(dlat, act=_bwd_act(ACT_RELU2), act_in=s['hpre'])
            blk['proj'].wgrad(dlat, s['hact'])
            blk['fc'].wgrad(dhpre, s['f'])
            df = blk['fc'].dgrad(dhpre)
            dlat1, _ = blk['ln_ff'].bwd(df, s['lat1'], s['m3'], s['r3'], dskip=dlat)
            do = blk['out'].dgrad(dlat1)
            blk['out'].wgrad(dlat1, s['o'])
            dq = torch.empty_like(s['q'])
            dkv = torch.empty_like(s['kv'])
            ks = (KV * 2 * W, 2 * W)
            ops.attention_bwd(do, s['q'], s['kv'][:, :W], s['kv'][:, W:], s['o'], s['lse'], B, H, L, KV, dh, q_strides=(L * W, W),
                              k_strides=ks, v_strides=ks, dq=dq, dk=dkv[:, :W], dv=dkv[:, W:], dq_strides=(L * W, W), dk_strides=ks,
                              dv_strides=ks)
            wq, wkv = P.w2(blk['inw'], W, W), P.w2(blk['inw'], 3 * W, W)[W:]
            gw = P.g2(blk['inw'], 3 * W, W)
            if gw is not None:
                gb = P.gvec(blk['inb'], 3 * W)

                ops.WQ.add_gemm(dq, s['qin'], gw[:W], W, W, B * L)
                ops.WQ.add_gemm(dkv, s['kvin'], gw[W:], 2 * W, W, B * KV)
                ops.WQ.add_colsum(dq, gb[:W], W)
                ops.WQ.add_colsum(dkv, gb[W:], 2 * W)
            dqin = ops.gemm(dq, wq, trans_b=True)
            dkvin = ops.gemm(dkv, wkv, trans_b=True)
            dlat, _ = blk['ln_1'].bwd(dqin, s['lat'], s['m1'], s['r1'], dy2=dkvin, dy2_map=RowMap(L, KV, 0), dskip=dlat1)
            last = blk is self.rblocks[0]
            dxf, _ = blk['ln_2'].bwd(dkvin, xf, s['m2'], s['r2'], dy_map=RowMap(Mx, KV, L), dskip=dxf,
                                     dx=(dxf_out if last else None))
        g = P.g('resampler.latents')
        if g is not None:                                                # sum over the batch of d(repeat(latents))
            ops.colsum(dlat.view(B, L * W), g.view(-1))
        return dxf

    # ---------------------------------------------------------------------------------------- ViT blocks
    def block_fwd(self, blk, x, B, S, sv_list):
        d = self.d
        W, H = d.width, d.vit_heads
        dh = W // H
        a, m1, r1 = blk['ln_1'].fwd(x)
        qkv = blk['qkv'].fwd(a)
        st = (S * 3 * W, 3 * W)
        o, lse = ops.attention_fwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], B, H, S, S, dh, q_strides=st, k_strides=st, v_strides=st)
        x1 = blk['out'].fwd(o, residual=x)
        b_, m2, r2 = blk['aln'].fwd(x1)                                   # Adaptor, pre-norm (utils.py:63-64)
        dpre = torch.empty_like(b_)
        dact = blk['down'].fwd(b_, act=ACT_RELU2, pre_out=dpre, pre_grad=SAVE_ACT_GRAD)
        x2 = blk['up'].fwd(dact, residual=x1)
        c, m3, r3 = blk['ln_2'].fwd(x2)
        fpre = torch.empty(x.shape[0], 4 * W, dtype=BF16, device=x.device)
        fact = blk['fc'].fwd(c, act=ACT_QUICKGELU, pre_out=fpre, pre_grad=SAVE_ACT_GRAD)
        x3 = blk['proj'].fwd(fact, residual=x2)
        if sv_list is not None:
            sv_list.append(dict(x=x, a=a, m1=m1, r1=r1, qkv=qkv, o=o, lse=lse, x1=x1, b=b_, m2=m2, r2=r2, dpre=dpre, dact=dact, x2=x2,
                                c=c, m3=m3, r3=r3, fpre=fpre, fact=fact))
        return x3

    def block_bwd(self, blk, s, dx3, B, S):
        d = self.d
        W, H = d.width, d.vit_heads
        dh = W // H
        dfpre = blk['proj'].dgrad(dx3, act=_bwd_act(ACT_QUICKGELU), act_in=s['fpre'])
        blk['proj'].wgrad(dx3, s['fact'])
        blk['fc'].wgrad(dfpre, s['c'])
        dc = blk['fc'].dgrad(dfpre)
        dx2, _ = blk['ln_2'].bwd(dc, s['x2'], s['m3'], s['r3'], dskip=dx3)
        ddpre = blk['up'].dgrad(dx2, act=_bwd_act(ACT_RELU2), act_in=s['dpre'])
        blk['up'].wgrad(dx2, s['dact'])
        blk['down'].wgrad(ddpre, s['b'])
        db = blk['down'].dgrad(ddpre)
        dx1, _ = blk['aln'].bwd(db, s['x1'], s['m2'], s['r2'], dskip=dx2)
        do = blk['out'].dgrad(dx1)
        blk['out'].wgrad(dx1, s['o'])
        dqkv = torch.empty_like(s['qkv'])
        st = (S * 3 * W, 3 * W)
        qkv = s['qkv']
        ops.attention_bwd(do, qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], s['o'], s['lse'], B, H, S, S, dh, q_strides=st, k_strides=st,
                          v_strides=st, dq=dqkv[:, :W], dk=dqkv[:, W:2 * W], dv=dqkv[:, 2 * W:], dq_strides=st, dk_strides=st, dv_strides=st)
        blk['qkv'].wgrad(dqkv, s['a'])
        da = blk['qkv'].dgrad(dqkv)
        dx, _ = blk['ln_1'].bwd(da, s['x'], s['m1'], s['r1'], dskip=dx1)
        return dx

    # ---------------------------------------------------------------------------------------- whole encoder
    # The encoder is split in two so that the Trainer can pipeline micro-batches:
    #   front  = patch embed + six expert stems (train-mode BatchNorm needs the WHOLE batch)  -> token buffers h, xf
    #   trunk  = resampler + ln_pre + ViT blocks + ln_post (sample-independent: any contiguous slice of the batch)
    @staticmethod
    def _instance_ids(val):
        """int64 [B,1,E,E] instance map of the obj_detection expert (dataset/utils.py:149: the raw label map); the compact input
        form carries it as the uint8 map itself"""
        if 'instance' in val:
            return val['instance'].contiguous()
        m = val['label_map']
        return (m if m.dim() == 4 else m.unsqueeze(1)).to(torch.int64).contiguous()

    def forward_front(self, x, inst_table, training, save):
        d, P = self.d, self.P
        W, p = d.width, d.patch_size
        rgb = x['rgb']
        B = rgb.shape[0]
        N, S = d.num_rgb_tokens, d.seq_len
        dev = rgb.device
        sv = {} if save else None
        h = torch.empty(B * S, W, dtype=BF16, device=dev)
        col = ops.patchify(rgb.contiguous().float(), p, self.Kp_rgb)
        shadow, _ = self.conv_shadow('conv1.rgb.weight', p)
        feat = ops.gemm(col, shadow)
        ops.tokens_finalize(feat, P.f('positional_embedding'), h, B, N, W, S, 0)
        names = [k for k in x if k != 'rgb']
        xf = None
        if names:
            G = d.expert_grid ** 2
            Mx = len(names) * G
            assert Mx == d.num_expert_tokens, 'expert dict does not match the configured experts'
            xf = torch.empty(B * Mx, W, dtype=BF16, device=dev)
            pos_e = self.expert_pos()
            keep = []
            feats = self.stems_fwd(x, names, training, sv)
            for ei, name in enumerate(names):
                val = x[name]
                f = feats[ei]
                if f.shape[0] != B * G:
                    raise RuntimeError(f'expert map {name}: stem produced {f.shape[0] // B} tokens per image, program expects {G} '
                                       f'(expert_resolution={d.expert_resolution})')
                if name == 'obj_detection':
                    inst = self._instance_ids(val)
                    ops.tokens_finalize(f, pos_e, xf, B, G, W, Mx, ei * G, inst, inst.shape[-1], d.expert_grid, inst_table,
                                        P.f('instance_embedding'))
                else:
                    ops.tokens_finalize(f, pos_e, xf, B, G, W, Mx, ei * G)
                keep.append(f)
            del keep
            if training and self._bn_flat is not None:
                self._advance_bn_counters(names)
        if save:
            sv.update(B=B, names=names, rgb_col=col,
                      inst=(self._instance_ids(x['obj_detection']) if 'obj_detection' in x else None), inst_table=inst_table)
        return h, xf, sv

    def _advance_bn_counters(self, names):
        """num_batches_tracked += 1 for the stems that ran (vit.py:86-120: only the experts present in the batch go through their conv1)"""
        lo, flat = self._bn_flat.data_ptr(), self._bn_flat
        slo, sflat = self._bn_stats_flat.data_ptr(), self._bn_stats_flat
        for m in (self._bn_mods[0], self._bn_mods[-1]):      # Module._apply (.to / .double) replaces ALL buffers: the views would silently stop advancing
            p, q = m.num_batches_tracked.data_ptr(), m.running_var.data_ptr()
            if not (lo <= p < lo + flat.numel() * 8) or not (slo <= q < slo + sflat.numel() * 4):
                raise RuntimeError('a BatchNorm counter no longer aliases the program\'s counter buffer (module moved or cast after the '
                                   'program was built): rebuild the program (module._prog = None)')
        # one advance per CALL of a stem: the reference runs conv1['seg'] once per seg_* expert of the batch (vit.py:136-139), so two seg experts advance
        # the shared stem's counters by two (round-5 advisor finding: they advanced by one)
        calls = {}
        for n in names:
            dom = 'seg' if 'seg' in n else n
            if dom in self._bn_range:
                calls[dom] = calls.get(dom, 0) + 1
        if len(calls) == len(self._bn_range) and all(c == 1 for c in calls.values()):
            ops.add_i64(flat, 1)
            return
        for dom, c in calls.items():
            a, n = self._bn_range[dom]
            if n:
                ops.add_i64(flat[a:a + n], c)

    def forward_trunk(self, h, xf, B, save):
        """h: [B*S, W] rows of B images (rgb tokens filled; the latent rows are written here); xf: [B*Mx, W] or None."""
        d = self.d
        S, W = d.seq_len, d.width
        sv = {} if save else None
        if xf is not None:
            self.resampler_fwd(xf, B, h, sv)
        h0, mp, rp = self.ln_pre.fwd(h)
        blocks_sv = [] if save else None
        t = h0
        for blk in self.blocks:
            t = self.block_fwd(blk, t, B, S, blocks_sv)
        out, mo, ro = self.ln_post.fwd(t)
        if save:
            sv.update(B=B, h=h, mp=mp, rp=rp, blocks=blocks_sv, t=t, mo=mo, ro=ro, has_x=xf is not None)
        return out.view(B, S, W), sv

    def backward_trunk(self, sv, dout, dh_out, dxf_out):
        """dout [B,S,W] -> dh_out [B*S, W] (gradient of the token buffer) and dxf_out [B*Mx, W] (expert tokens), both given
        as (slices of) preallocated buffers."""
        d = self.d
        W = d.width
        B, S, N = sv['B'], d.seq_len, d.num_rgb_tokens
        dt, _ = self.ln_post.bwd(dout.reshape(B * S, W), sv['t'], sv['mo'], sv['ro'])
        for blk, s in zip(reversed(self.blocks), reversed(sv['blocks'])):
            dt = self.block_bwd(blk, s, dt, B, S)
        self.ln_pre.bwd(dt, sv['h'], sv['mp'], sv['rp'], dx=dh_out)
        if sv['has_x']:
            L = d.num_latents
            dlat = torch.empty(B * L, W, dtype=BF16, device=dt.device)
            ops.copy_rows(dh_out, dlat, B * L, W, src_map=RowMap(L, S, N))
            self.resampler_bwd(dlat, B, sv, dxf_out)

    def backward_front(self, sv, dh, dxf):
        d, P = self.d, self.P
        W = d.width
        B, S, N = sv['B'], d.seq_len, d.num_rgb_tokens
        gpos = P.g('positional_embedding')
        names = sv['names']
        if names:
            G = d.expert_grid ** 2
            Mx = len(names) * G
            same = d.expert_grid == d.rgb_grid
            dpos_e = gpos if same else ops.zeros((G, W), F32, dh.device)
            keep = []
            for ei, name in enumerate(names):
                dfeat = torch.empty(B * G, W, dtype=BF16, device=dh.device)
                if name == 'obj_detection':
                    inst = sv['inst']
                    ops.tokens_finalize_bwd(dxf, dfeat, dpos_e, B, G, W, Mx, ei * G, inst, inst.shape[-1], d.expert_grid, sv['inst_table'],
                                            P.g('instance_embedding'))
                else:
                    ops.tokens_finalize_bwd(dxf, dfeat, dpos_e, B, G, W, Mx, ei * G)
                keep.append(dfeat)
            self.stems_bwd(['seg' if 'seg' in n else n for n in names], keep, sv)
            del keep
            if not same and gpos is not None:
                self.expert_pos_bwd(dpos_e)
        drgb = torch.empty(B * N, W, dtype=BF16, device=dh.device)
        ops.tokens_finalize_bwd(dh, drgb, gpos, B, N, W, S, 0)
        self.conv_wgrad('conv1.rgb.weight', d.patch_size, drgb, sv['rgb_col'])

    def forward(self, x, inst_table, training, save):
        h, xf, svf = self.forward_front(x, inst_table, training, save)
        out, svt = self.forward_trunk(h, xf, x['rgb'].shape[0], save)
        return out, ((svf, svt) if save else None)

    def backward(self, sv, dout):
        """dout: [B, S, W] bf16 (batch-major). Accumulates every trainable gradient into the store's buffer."""
        svf, svt = sv
        d = self.d
        B = svt['B']
        dh = torch.empty(B * d.seq_len, d.width, dtype=BF16, device=dout.device)
        dxf = torch.empty(B * d.num_expert_tokens, d.width, dtype=BF16, device=dout.device) if svt['has_x'] else None
        self.backward_trunk(svt, dout, dh, dxf)
        self.backward_front(svf, dh, dxf)
