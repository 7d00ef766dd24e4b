"""Layer programs (forward + hand-written backward) of the Prismer language decoder on the HIP operator set.

Reference graph: RobertaForCausalLMModified.forward (model/modules/roberta.py:358-399) =
RobertaEmbeddings (:66-76) -> 12 x [self-attn, cross-attn, Adaptor(norm_late), MLP] + output_layer (:223-231)
-> RobertaLMHead (:421-426) -> shifted label-smoothed CE (:381-387).
Differences in FORM (not in arithmetic): the three q/k/v nn.Linear of a self-attention run as one packed GEMM
(their parameters are adjacent in the flat store), cross-attention k/v likewise; every
`LayerNorm(dropout(dense(x)) + residual)` is a GEMM with a fused bias+dropout+residual epilogue followed by the
LayerNorm kernel; the post-LN residual stream (LayerNorm outputs and the pre-norm sums) is kept in fp32 next to the bf16
copies the GEMMs read -- what autocast does on the reference (its fp32 LayerNorm returns fp32 to an fp32 residual); logits live in a [B*T, Vpad] bf16 buffer (Vpad = vocab rounded up to 64) and the backward
overwrites it with dlogits.
"""
import torch

from .. import ops
from .._lib import ACT_GELU, ACT_RELU2
from .encoder import LN, SAVE_ACT_GRAD, Linear, _bwd_act

BF16, F32 = torch.bfloat16, torch.float32


class DecoderProgram:
    def __init__(self, module, dims, store):
        self.mod, self.d, self.P = module, dims, store
        d, P = dims, store
        H, Hv = d.hidden_size, d.vision_hidden_size
        eps = d.layer_norm_eps
        self.Vpad = (d.vocab_size + 63) // 64 * 64

        def self_attn(p):
            return dict(qkv=Linear(P, p + 'self.query.weight', p + 'self.query.bias', rows=3 * H, cols=H),
                        out=Linear(P, p + 'output.dense.weight', p + 'output.dense.bias'), ln=LN(P, p + 'output.LayerNorm', eps))

        def mlp(p):
            return dict(inter=Linear(P, p + 'intermediate.dense.weight', p + 'intermediate.dense.bias'),
                        out=Linear(P, p + 'output.dense.weight', p + 'output.dense.bias'), ln=LN(P, p + 'output.LayerNorm', eps))

        self.layers = []
        for l in range(d.num_hidden_layers):
            p = f'roberta.encoder.layer.{l}.'
            self.layers.append(dict(
                idx=l, sa=self_attn(p + '0.attention.'), mlp=mlp(p + '0.'),
                ca=dict(q=Linear(P, p + '1.self.query.weight', p + '1.self.query.bias'),
                        kv=Linear(P, p + '1.self.key.weight', p + '1.self.key.bias', rows=2 * H, cols=Hv),
                        out=Linear(P, p + '1.output.dense.weight', p + '1.output.dense.bias'), ln=LN(P, p + '1.output.LayerNorm', eps)),
                ad=dict(down=Linear(P, p + '2.adaptor.down_proj.weight', p + '2.adaptor.down_proj.bias'),
                        up=Linear(P, p + '2.adaptor.up_proj.weight', p + '2.adaptor.up_proj.bias'), ln=LN(P, p + '2.adaptor_ln', 1e-5))))
        # all cross-attention K/V projections read the same encoder output: when the store placed them back-to-back
        # (store._reorder_qkv) they are ONE linear of 2*H*layers output features -- forward, dgrad and wgrad
        self.kv_all = None
        if self.layers:
            L = len(self.layers)
            k0 = 'roberta.encoder.layer.0.1.self.key.'
            o = P.offset
            ok = all(o[f'roberta.encoder.layer.{l}.1.self.key.weight'] - o[k0 + 'weight'] == 2 * l * H * Hv and
                     o[f'roberta.encoder.layer.{l}.1.self.value.weight'] - o[k0 + 'weight'] == (2 * l + 1) * H * Hv and
                     o[f'roberta.encoder.layer.{l}.1.self.key.bias'] - o[k0 + 'bias'] == 2 * l * H and
                     o[f'roberta.encoder.layer.{l}.1.self.value.bias'] - o[k0 + 'bias'] == (2 * l + 1) * H for l in range(L))
            same = len({P.is_trainable(f'roberta.encoder.layer.{l}.1.self.{w}.{t}') for l in range(L) for w in ('key', 'value')
                        for t in ('weight', 'bias')}) == 1
            if ok and same:
                self.kv_all = Linear(P, k0 + 'weight', k0 + 'bias', rows=2 * H * L, cols=Hv)
        p = 'roberta.encoder.output_layer.'
        self.final = dict(idx=d.num_hidden_layers, sa=self_attn(p + 'attention.'), mlp=mlp(p))
        self.head_dense = Linear(P, 'lm_head.dense.weight', 'lm_head.dense.bias')
        self.head_ln = LN(P, 'lm_head.layer_norm', eps)
        self.emb_ln = LN(P, 'roberta.embeddings.LayerNorm', eps)
        # the q/k/v packing relies on adjacency in the flat store
        o = P.offset
        n = 'roberta.encoder.output_layer.attention.self.'
        assert o[n + 'key.weight'] - o[n + 'query.weight'] == H * H and o[n + 'value.weight'] - o[n + 'key.weight'] == H * H
        assert o[n + 'key.bias'] - o[n + 'query.bias'] == H, 'q/k/v biases must be adjacent (H % 64 == 0 required)'

    site_base = 0      # added to every dropout call-site id: the Trainer gives each micro-batch its own Philox streams
    kv_prefetch = True # pre-issue the cross-attention K/V projections on a branch stream (off inside micro-batch branches)

    def drop(self, site, p, seed):
        return ops.Dropout(p, seed, self.site_base + site) if (seed is not None and p > 0.0) else None

    # ---------------------------------------------------------------------------------------- sub-blocks
    def post_ln(self, ln, s):
        """LayerNorm of the fp32 pre-norm sum -> (bf16 copy for the next GEMMs, fp32 copy = residual stream)"""
        yf = torch.empty_like(s)
        y, m, r = ln.fwd(s, out_f32=yf)
        return y, yf, m, r

    def self_attn_fwd(self, blk, li, h, hf, B, T, key_mask, seed, sv):
        d = self.d
        H, nh = d.hidden_size, d.num_attention_heads
        dh = H // nh
        qkv = blk['qkv'].fwd(h)
        st = (T * 3 * H, 3 * H)
        dr_a = self.drop(li * 16 + 1, d.attention_probs_dropout_prob, seed)
        o, lse = ops.attention_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, T, T, dh, q_strides=st, k_strides=st, v_strides=st,
                                   key_mask=key_mask, causal=True, drop=dr_a)
        dr_h = self.drop(li * 16 + 2, d.hidden_dropout_prob, seed)
        s = blk['out'].fwd(o, drop=dr_h, residual=hf, out_f32=True)
        y, yf, m, r = self.post_ln(blk['ln'], s)
        if sv is not None:
            sv.append(dict(h=h, qkv=qkv, o=o, lse=lse, s=s, m=m, r=r, dr_a=dr_a, dr_h=dr_h))
        return y, yf

    def self_attn_bwd(self, blk, s, dy, B, T, key_mask):
        d = self.d
        H, nh = d.hidden_size, d.num_attention_heads
        dh = H // nh
        ds, dsd = blk['ln'].bwd(dy, s['s'], s['m'], s['r'], drop=s['dr_h'])
        do = blk['out'].dgrad(dsd)
        blk['out'].wgrad(dsd, s['o'])
        qkv = s['qkv']
        dqkv = torch.empty_like(qkv)
        st = (T * 3 * H, 3 * H)
        ops.attention_bwd(do, qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], s['o'], s['lse'], B, nh, T, T, dh, q_strides=st, k_strides=st,
                          v_strides=st, dq=dqkv[:, :H], dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:], dq_strides=st, dk_strides=st, dv_strides=st,
                          key_mask=key_mask, causal=True, drop=s['dr_a'])
        blk['qkv'].wgrad(dqkv, s['h'])
        return blk['qkv'].dgrad(dqkv, residual=ds)

    def cross_attn_fwd(self, blk, li, h, hf, enc, B, T, S, seed, sv, kv=None, kv_ready=None):
        d = self.d
        H, nh = d.hidden_size, d.num_attention_heads
        dh = H // nh
        q = blk['q'].fwd(h)
        if kv is None:
            kv = blk['kv'].fwd(enc)
        elif kv_ready is not None:
            torch.cuda.current_stream().wait_event(kv_ready)       # the K/V projections ran on the branch stream
        ks = (S * kv.stride(0), kv.stride(0))                      # kv may be a column slice of the all-layers buffer
        dr_a = self.drop(li * 16 + 3, d.attention_probs_dropout_prob, seed)
        o, lse = ops.attention_fwd(q, kv[:, :H], kv[:, H:], B, nh, T, S, dh, q_strides=(T * H, H), k_strides=ks, v_strides=ks, drop=dr_a)
        dr_h = self.drop(li * 16 + 4, d.hidden_dropout_prob, seed)
        s = blk['out'].fwd(o, drop=dr_h, residual=hf, out_f32=True)
        y, yf, m, r = self.post_ln(blk['ln'], s)
        if sv is not None:
            sv.append(dict(h=h, q=q, kv=kv, o=o, lse=lse, s=s, m=m, r=r, dr_a=dr_a, dr_h=dr_h))
        return y, yf

    def cross_attn_bwd(self, blk, s, dy, enc, denc, B, T, S, dkv=None):
        """dkv: this layer's column slice of the all-layers dK/dV buffer (merged projection: its dgrad / wgrad run once,
        after the last layer); None: per-layer K/V linear, d(enc) accumulated into the fp32 buffer `denc`."""
        d = self.d
        H, nh = d.hidden_size, d.num_attention_heads
        dh = H // nh
        ds, dsd = blk['ln'].bwd(dy, s['s'], s['m'], s['r'], drop=s['dr_h'])
        do = blk['out'].dgrad(dsd)
        blk['out'].wgrad(dsd, s['o'])
        dq = torch.empty_like(s['q'])
        merged = dkv is not None
        if not merged:
            dkv = torch.empty(s['kv'].shape, dtype=BF16, device=dy.device)
        kv = s['kv']
        ks, dks = (S * kv.stride(0), kv.stride(0)), (S * dkv.stride(0), dkv.stride(0))
        ops.attention_bwd(do, s['q'], kv[:, :H], kv[:, H:], s['o'], s['lse'], B, nh, T, S, dh, q_strides=(T * H, H), k_strides=ks,
                          v_strides=ks, dq=dq, dk=dkv[:, :H], dv=dkv[:, H:], dq_strides=(T * H, H), dk_strides=dks, dv_strides=dks,
                          drop=s['dr_a'])
        blk['q'].wgrad(dq, s['h'])
        if not merged:
            blk['kv'].wgrad(dkv, enc)
            # d(enc) is only needed after the last layer: its accumulating GEMMs leave the critical path (fp32 accumulate,
            # serialised on the side stream)
            ops.off_critical_path(lambda: blk['kv'].dgrad(dkv, out=denc, out_f32=True, accumulate=True), dkv)
        return blk['q'].dgrad(dq, residual=ds)

    def adaptor_fwd(self, blk, h, hf, sv):
        dpre = torch.empty_like(h)
        dact = blk['down'].fwd(h, act=ACT_RELU2, pre_out=dpre, pre_grad=SAVE_ACT_GRAD)
        s = blk['up'].fwd(dact, residual=hf, out_f32=True)
        y, yf, m, r = self.post_ln(blk['ln'], s)                          # norm_late (utils.py:61-62)
        if sv is not None:
            sv.append(dict(h=h, dpre=dpre, dact=dact, s=s, m=m, r=r))
        return y, yf

    def adaptor_bwd(self, blk, s, dy):
        ds, _ = blk['ln'].bwd(dy, s['s'], s['m'], s['r'])
        ddpre = blk['up'].dgrad(ds, act=_bwd_act(ACT_RELU2), act_in=s['dpre'])
        blk['up'].wgrad(ds, s['dact'])
        blk['down'].wgrad(ddpre, s['h'])
        return blk['down'].dgrad(ddpre, residual=ds)

    def mlp_fwd(self, blk, li, h, hf, seed, sv):
        d = self.d
        ipre = torch.empty(h.shape[0], d.intermediate_size, dtype=BF16, device=h.device)
        iact = blk['inter'].fwd(h, act=ACT_GELU, pre_out=ipre, pre_grad=SAVE_ACT_GRAD)
        dr_h = self.drop(li * 16 + 5, d.hidden_dropout_prob, seed)
        s = blk['out'].fwd(iact, drop=dr_h, residual=hf, out_f32=True)
        y, yf, m, r = self.post_ln(blk['ln'], s)
        if sv is not None:
            sv.append(dict(h=h, ipre=ipre, iact=iact, s=s, m=m, r=r, dr_h=dr_h))
        return y, yf

    def mlp_bwd(self, blk, s, dy):
        ds, dsd = blk['ln'].bwd(dy, s['s'], s['m'], s['r'], drop=s['dr_h'])
        dipre = blk['out'].dgrad(dsd, act=_bwd_act(ACT_GELU), act_in=s['ipre'])
        blk['out'].wgrad(dsd, s['iact'])
        blk['inter'].wgrad(dipre, s['h'])
        return blk['inter'].dgrad(dipre, residual=ds)

    # ---------------------------------------------------------------------------------------- whole decoder
    def forward(self, input_ids, attention_mask, enc, labels, seed, save, want_logits=True):
        """enc: [B, S, Hv] bf16 contiguous. seed: int64[1] device tensor or None (no dropout = eval mode).
        Returns (logits_buf [B*T, Vpad] bf16, loss [B] fp32 or None, saved)."""
        d, P = self.d, self.P
        B, T = input_ids.shape
        S = enc.shape[1]
        H = d.hidden_size
        e = 'roberta.embeddings.'
        input_ids = input_ids.contiguous()
        key_mask = None
        if attention_mask is not None:
            key_mask = (attention_mask != 0).to(torch.uint8).contiguous()
        enc2 = enc.reshape(B * S, enc.shape[2])
        sv = [] if save else None
        dr_e = self.drop(9000, d.hidden_dropout_prob, seed)
        h, xhat, erstd, hf = ops.embed_fwd(input_ids, P.f(e + 'word_embeddings.weight'), P.f(e + 'position_embeddings.weight'),
                                       P.f(e + 'token_type_embeddings.weight'), P.f(e + 'LayerNorm.weight'), P.f(e + 'LayerNorm.bias'),
                                       d.layer_norm_eps, d.pad_token_id, dr_e, want_f32=True)
        # the K/V projections of all cross-attention layers depend on `enc` only: issue them up front on a branch stream
        # (12 x [B*S, 2H] GEMMs that overlap the latency-bound decoder chain); each layer waits on its own event.
        kvs, kv_evs = [None] * len(self.layers), [None] * len(self.layers)
        branch = self.kv_prefetch and isinstance(ops.POOL, ops.BranchPool)
        if self.kv_all is not None:                      # ONE [B*S, 2H*layers] GEMM; every layer reads its column slice
            def all_kv():
                return self.kv_all.fwd(enc2)
            if branch:
                with ops.POOL.branch(0):
                    kva = all_kv()
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                ops.POOL.used.clear()                    # joined through the event the first cross-attention waits on
                kv_evs[0] = ev
            else:
                kva = all_kv()
            kvs = [kva[:, 2 * H * i:2 * H * (i + 1)] for i in range(len(self.layers))]
        elif branch:
            with ops.POOL.branch(0):
                for i, L in enumerate(self.layers):
                    kvs[i] = L['ca']['kv'].fwd(enc2)
                    kv_evs[i] = torch.cuda.Event()
                    kv_evs[i].record(torch.cuda.current_stream())
            ops.POOL.used.clear()        # joined through the per-layer events below (the last one covers the whole branch)
        for i, L in enumerate(self.layers):
            h, hf = self.self_attn_fwd(L['sa'], L['idx'], h, hf, B, T, key_mask, seed, sv)
            h, hf = self.cross_attn_fwd(L['ca'], L['idx'], h, hf, enc2, B, T, S, seed, sv, kvs[i], kv_evs[i])
            h, hf = self.adaptor_fwd(L['ad'], h, hf, sv)
            h, hf = self.mlp_fwd(L['mlp'], L['idx'], h, hf, seed, sv)
        F_ = self.final
        h, hf = self.self_attn_fwd(F_['sa'], F_['idx'], h, hf, B, T, key_mask, seed, sv)
        h, hf = self.mlp_fwd(F_['mlp'], F_['idx'], h, hf, seed, sv)
        t0pre = torch.empty_like(h)
        t0 = self.head_dense.fwd(h, act=ACT_GELU, pre_out=t0pre)
        t1, hm, hr = self.head_ln.fwd(t0)
        V = d.vocab_size
        logits = torch.empty(B * T, self.Vpad, dtype=BF16, device=h.device)
        ops.gemm(t1, P.w(e + 'word_embeddings.weight'), out=logits, bias=P.f('lm_head.bias'), N=V)      # tied decoder weight
        loss = row_lse = None
        if labels is not None:
            labels = labels.contiguous()
            loss, row_lse = ops.ce_fwd(logits, labels, B, T, V, d.label_smoothing)
        saved = None
        if save:
            saved = dict(blocks=sv, B=B, T=T, S=S, ids=input_ids, key_mask=key_mask, enc=enc2, xhat=xhat, erstd=erstd, dr_e=dr_e, hL=h,
                         t0pre=t0pre, t0=t0, t1=t1, hm=hm, hr=hr, logits=logits, labels=labels, row_lse=row_lse)
        return logits, loss, saved

    # ---------------------------------------------------------------------------------------- KV-cached inference
    # The reference's generate (prismer_caption.py:45-50, prismer_vqa.py:52-58 through roberta.py:401-406) re-runs the whole
    # prefix and re-projects the encoder output at every step.  Here the cross-attention K/V of all layers are projected ONCE per
    # image batch (the merged projection of the training path) and every self-attention layer keeps its K/V rows in a cache
    # [layers+1, rows, Tmax, 2H]; a step runs the new position(s) only.
    def decode_begin(self, enc, Tmax):
        """enc: [Bb, S, Hv] bf16 (already expanded to one row per beam).  Returns the decoding state."""
        d = self.d
        Bb, S, Hv = enc.shape
        H = d.hidden_size
        enc2 = enc.reshape(Bb * S, Hv)
        if self.kv_all is not None:
            kva = self.kv_all.fwd(enc2)
            kvs = [kva[:, 2 * H * i:2 * H * (i + 1)] for i in range(len(self.layers))]
        else:
            kvs = [L['ca']['kv'].fwd(enc2) for L in self.layers]
        nl = len(self.layers) + 1
        cache = [torch.zeros(nl, Bb, Tmax, 2 * H, dtype=BF16, device=enc.device) for _ in range(2)]     # ping-pong for beam reordering
        return dict(kvs=kvs, cache=cache, cur=0, Bb=Bb, S=S, Tmax=Tmax, filled=0)

    def decode_reorder(self, st, rows):
        """beam step: hypothesis r continues former hypothesis rows[r] (int32 device tensor [Bb]) -- caches follow"""
        src, dst = st['cache'][st['cur']], st['cache'][st['cur'] ^ 1]
        nl, Bb, Tmax, W2 = src.shape
        idx = (torch.arange(nl, device=rows.device, dtype=torch.int32)[:, None] * Bb + rows[None, :].to(torch.int32)).reshape(-1).contiguous()
        ops.gather_rows(src.view(nl * Bb, Tmax * W2), idx, dst.view(nl * Bb, Tmax * W2), cols=st['filled'] * W2)
        st['cur'] ^= 1

    def _self_attn_cached(self, blk, li, slot, h, hf, st, t0, T, key_mask):
        d = self.d
        H, nh = d.hidden_size, d.num_attention_heads
        dh = H // nh
        Bb, Tmax = st['Bb'], st['Tmax']
        Sq = T - t0
        qkv = blk['qkv'].fwd(h)                                                  # [Bb*Sq, 3H]
        cache = st['cache'][st['cur']][slot]                                     # [Bb, Tmax, 2H]
        # K | V rows of the new positions -> cache[:, t0:T]
        ops.copy_rows(qkv[:, H:], cache.view(Bb * Tmax, 2 * H), Bb * Sq, 2 * H, dst_map=ops.RowMap(Sq, Tmax, t0), src_ld=3 * H, dst_ld=2 * H)
        cs = (Tmax * 2 * H, 2 * H)
        o, _ = ops.attention_fwd(qkv[:, :H], cache[:, :, :H], cache[:, :, H:], Bb, nh, Sq, T, dh, q_strides=(Sq * 3 * H, 3 * H), k_strides=cs,
                                 v_strides=cs, key_mask=key_mask, causal=Sq > 1)
        s = blk['out'].fwd(o, residual=hf, out_f32=True)
        y, yf, _, _ = self.post_ln(blk['ln'], s)
        return y, yf

    def decode(self, st, input_ids, attention_mask, t0):
        """runs positions t0 .. T-1 of input_ids [Bb, T] (t0 == 0: prefill, causal; otherwise exactly the last position) against
        the caches and returns the logits of the LAST position, fp32 view [Bb, V] of a bf16 buffer."""
        d, P = self.d, self.P
        Bb, T = input_ids.shape
        assert Bb == st['Bb'] and T <= st['Tmax'] and (t0 == 0 or t0 == T - 1), 'decode: prefill (t0=0) or single-token steps only'
        H, S = d.hidden_size, st['S']
        e = 'roberta.embeddings.'
        Sq = T - t0
        key_mask = None
        if attention_mask is not None:
            key_mask = (attention_mask != 0).to(torch.uint8).contiguous()
        h_all, _, _, hf_all = ops.embed_fwd(input_ids.contiguous(), P.f(e + 'word_embeddings.weight'), P.f(e + 'position_embeddings.weight'),
                                            P.f(e + 'token_type_embeddings.weight'), P.f(e + 'LayerNorm.weight'), P.f(e + 'LayerNorm.bias'),
                                            d.layer_norm_eps, d.pad_token_id, None, want_f32=True)
        if t0 == 0:
            h, hf = h_all, hf_all
        else:                                                                     # the new position of every row
            h = h_all.view(Bb, T, H)[:, t0:].reshape(Bb * Sq, H).contiguous()
            hf = hf_all.view(Bb, T, H)[:, t0:].reshape(Bb * Sq, H).contiguous()
        for i, L in enumerate(self.layers):
            h, hf = self._self_attn_cached(L['sa'], L['idx'], i, h, hf, st, t0, T, key_mask)
            h, hf = self.cross_attn_fwd(L['ca'], L['idx'], h, hf, None, Bb, Sq, S, None, None, st['kvs'][i])
            h, hf = self.adaptor_fwd(L['ad'], h, hf, None)
            h, hf = self.mlp_fwd(L['mlp'], L['idx'], h, hf, None, None)
        F_ = self.final
        h, hf = self._self_attn_cached(F_['sa'], F_['idx'], len(self.layers), h, hf, st, t0, T, key_mask)
        h, hf = self.mlp_fwd(F_['mlp'], F_['idx'], h, hf, None, None)
        st['filled'] = max(st['filled'], T)
        last = h if Sq == 1 else h.view(Bb, Sq, H)[:, -1].contiguous()           # LM head on the last position only
        t0_ = self.head_dense.fwd(last, act=ACT_GELU)
        t1, _, _ = self.head_ln.fwd(t0_, save_stats=False)
        V = d.vocab_size
        logits = torch.empty(Bb, self.Vpad, dtype=BF16, device=h.device)
        ops.gemm(t1, P.w(e + 'word_embeddings.weight'), out=logits, bias=P.f('lm_head.bias'), N=V)
        return logits[:, :V]

    def backward(self, sv, dloss):
        """dloss: fp32 [B] gradient of the per-sample losses. Returns d(enc) as bf16 [B, S, Hv]."""
        st = self.backward_start(sv, dloss)
        self.backward_layers(st, len(self.layers), 0)
        return self.backward_finish(st)

    # The backward is resumable so that the Trainer can cut it into hipGraph segments in reverse layer order: after each
    # segment the gradients of the layers it covered are complete (their flat-buffer range can go to the all-reduce while the
    # next segment runs).  backward_start: CE, LM head, output_layer; backward_layers(hi, lo): layers hi-1 .. lo;
    # backward_finish: embeddings + the merged cross-attention K/V projection (whose weights sit in layer 0's range).
    def backward_start(self, sv, dloss):
        d, P = self.d, self.P
        B, T, S = sv['B'], sv['T'], sv['S']
        V = d.vocab_size
        e = 'roberta.embeddings.'
        dlogits = ops.ce_bwd(sv['logits'], sv['labels'], B, T, V, d.label_smoothing, sv['row_lse'], dloss.contiguous())
        wname = e + 'word_embeddings.weight'
        gw = P.g(wname)
        gb = P.g('lm_head.bias')

        def head_wgrad():
            if gw is not None:
                ops.gemm(dlogits, sv['t1'], out=gw, trans_a=True, trans_b=True, out_f32=True, accumulate=True, M=V, N=d.hidden_size, K=B * T)
            if gb is not None:
                ops.colsum(dlogits, gb, N=V)
        ops.off_critical_path(head_wgrad, dlogits, sv['t1'])
        dt1 = ops.gemm(dlogits, P.w(wname), trans_b=True, K=V)            # [B*T, H]
        dt0, _ = self.head_ln.bwd(dt1, sv['t0'], sv['hm'], sv['hr'])
        dt0pre = ops.act_bwd(dt0, sv['t0pre'], ACT_GELU)
        self.head_dense.wgrad(dt0pre, sv['hL'])
        dh = self.head_dense.dgrad(dt0pre)
        blocks = list(sv['blocks'])
        H = d.hidden_size
        nl = len(self.layers)
        merged = self.kv_all is not None and nl > 0
        denc = None if merged else ops.zeros((B * S, d.vision_hidden_size), F32, dh.device)
        dkv_all = torch.empty(B * S, 2 * H * nl, dtype=BF16, device=dh.device) if merged else None
        F_ = self.final
        dh = self.mlp_bwd(F_['mlp'], blocks.pop(), dh)
        dh = self.self_attn_bwd(F_['sa'], blocks.pop(), dh, B, T, sv['key_mask'])
        return dict(sv=sv, dh=dh, blocks=blocks, merged=merged, denc=denc, dkv_all=dkv_all)

    def backward_layers(self, st, hi, lo):
        sv, d = st['sv'], self.d
        B, T, S, H = sv['B'], sv['T'], sv['S'], d.hidden_size
        dh, blocks, merged = st['dh'], st['blocks'], st['merged']
        for li in range(hi - 1, lo - 1, -1):
            L = self.layers[li]
            dh = self.mlp_bwd(L['mlp'], blocks.pop(), dh)
            dh = self.adaptor_bwd(L['ad'], blocks.pop(), dh)
            dh = self.cross_attn_bwd(L['ca'], blocks.pop(), dh, sv['enc'], st['denc'], B, T, S,
                                     dkv=st['dkv_all'][:, 2 * H * li:2 * H * (li + 1)] if merged else None)
            dh = self.self_attn_bwd(L['sa'], blocks.pop(), dh, B, T, sv['key_mask'])
        st['dh'] = dh

    def backward_finish(self, st):
        d, P = self.d, self.P
        sv, dh, merged, denc, dkv_all = st['sv'], st['dh'], st['merged'], st['denc'], st['dkv_all']
        B, S = sv['B'], sv['S']
        e = 'roberta.embeddings.'
        wname = e + 'word_embeddings.weight'
        # The tied word-embedding gradient has two writers, the LM-head wgrad GEMM (read-modify-write tiles) and these
        # scatter-adds: both live on the side stream, whose order serialises them (also across micro-batches).
        ops.off_critical_path(lambda: ops.embed_bwd(
            dh, sv['ids'], P.f(wname), P.f(e + 'position_embeddings.weight'), P.f(e + 'token_type_embeddings.weight'),
            P.f(e + 'LayerNorm.weight'), P.f(e + 'LayerNorm.bias'), d.layer_norm_eps, d.pad_token_id, sv['xhat'], sv['erstd'],
            sv['dr_e'], P.g(wname), P.g(e + 'position_embeddings.weight'), P.g(e + 'token_type_embeddings.weight'),
            P.g(e + 'LayerNorm.weight'), P.g(e + 'LayerNorm.bias')), dh, sv['xhat'], sv['erstd'], sv['ids'])
        # denc was accumulated on the side stream: its bf16 copy is produced there too.  The result is valid after the
        # caller's ops.join_side() -- this program never makes its own stream wait for the side stream (under stream capture
        # a forked stream that re-joins work it forked itself crashes this ROCm's capture_end; only the origin may join).
        denc_b = torch.empty(B * S, d.vision_hidden_size, dtype=BF16, device=dh.device)
        if merged:               # d(enc) = dKV_all . W_all (K = 2H*layers: fp32 accumulation inside ONE GEMM), dW_all = dKV_all^T enc
            self.kv_all.wgrad(dkv_all, sv['enc'])
            ops.off_critical_path(lambda: self.kv_all.dgrad(dkv_all, out=denc_b), dkv_all, denc_b)
        else:
            ops.off_critical_path(lambda: ops.cast_to_bf16(denc, out=denc_b), denc, denc_b)
        return denc_b.view(B, S, d.vision_hidden_size)
