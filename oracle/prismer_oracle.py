"""CPU oracle for the Prismer hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32 or fp64) functional restatement of the reference's forward algorithm for the
north-star path: VisionTransformer (+ expert stems, instance embedding, positional embedding, Experts
Resampler), RobertaForCausalLMModified (+ LM head and shifted label-smoothed CE) and the caption / VQA
training losses.  It works on a flat state dict with the reference's key names, so the
same weights can be fed to the reference module classes, to this oracle and to the HIP path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (prismer_amd/) never does.

Pinning: the reference ships no tests / golden vectors for this path, so the oracle is pinned
against OUTPUTS OF THE REFERENCE ITSELF: tests/golden/make_golden.py imports the reference module classes
from /root/reference in the build container, feeds them the synthetic weights/inputs of
prismer_amd/synth.py and commits the outputs under tests/golden/; tests/test_oracle_golden.py checks this
file against those fixtures (and against the live reference when /root/reference is present).

Training-mode dropout (round 5): the decoder functions take `drop(site, x)`; the masks are the caller's.  Pinned the same way: with the
reference's nn.Dropout modules replaced by the same callable the reference classes and this file agree to fp32 round-off
(tests/test_oracle_golden.py::test_oracle_dropout_sites_match_reference_with_identical_masks), and the committed fixtures
tests/golden/*_drop.npz hold the reference's outputs under the masks the HIP library draws for a fixed seed.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LABEL_DOMAINS = ('seg', 'obj_detection', 'ocr_detection')


# ------------------------------------------------------------------ small pieces

def layer_norm(x, w, b, eps=1e-5):
    """model/modules/utils.py:14-19 -- always computed in fp32 (fp64 stays fp64), cast back."""
    ct = torch.float64 if x.dtype == torch.float64 else torch.float32
    y = F.layer_norm(x.to(ct), (x.shape[-1],), w.to(ct), b.to(ct), eps)
    return y.to(x.dtype)


def quick_gelu(x):          # utils.py:23-25
    return x * torch.sigmoid(1.702 * x)


def squared_relu(x):        # utils.py:28-30
    return torch.relu(x) ** 2


def gelu_erf(x):            # transformers ACT2FN['gelu'] (roberta.py:164,423)
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x, sd, p):
    return F.linear(x, sd[p + '.weight'], sd[p + '.bias'])


def interpolate_pos_embed(pos, target_len):
    """utils.py:34-44: bicubic (align_corners=False) re-grid when the grids differ."""
    o = int(pos.shape[0] ** 0.5)
    n = int(target_len ** 0.5)
    if o == n:
        return pos
    g = pos.reshape(1, o, o, -1).permute(0, 3, 1, 2)
    g = F.interpolate(g, size=(n, n), mode='bicubic', align_corners=False)
    return g.permute(0, 2, 3, 1).flatten(0, 2)


def packed_mha(q_in, kv_in, in_w, in_b, out_w, out_b, heads):
    """nn.MultiheadAttention (vit.py:41,52-53; resampler.py:18,30-31), batch-first here.
    q_in [B,Lq,D], kv_in [B,Lk,D]; packed in-proj is q|k|v (torch _in_projection_packed)."""
    B, Lq, D = q_in.shape
    Lk = kv_in.shape[1]
    dh = D // heads
    q = F.linear(q_in, in_w[:D], in_b[:D])
    k = F.linear(kv_in, in_w[D:2 * D], in_b[D:2 * D])
    v = F.linear(kv_in, in_w[2 * D:], in_b[2 * D:])
    q = q.reshape(B, Lq, heads, dh).transpose(1, 2)
    k = k.reshape(B, Lk, heads, dh).transpose(1, 2)
    v = v.reshape(B, Lk, heads, dh).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, D)
    return F.linear(o, out_w, out_b)


def adaptor(x, sd, p, norm_late):
    """utils.py:48-65."""
    def mlp(t):
        return linear(squared_relu(linear(t, sd, p + 'adaptor.down_proj')), sd, p + 'adaptor.up_proj')
    if norm_late:
        return layer_norm(mlp(x) + x, sd[p + 'adaptor_ln.weight'], sd[p + 'adaptor_ln.bias'])
    return mlp(layer_norm(x, sd[p + 'adaptor_ln.weight'], sd[p + 'adaptor_ln.bias'])) + x


# ------------------------------------------------------------------ encoder front-end

def batch_norm(x, sd, p, train, bn_updates, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d (vit.py:91-119): train => biased batch var for normalisation, unbiased for the
    running update. bn_updates collects the new running stats (not applied in place)."""
    w, b = sd[p + '.weight'], sd[p + '.bias']
    rm, rv = sd[p + '.running_mean'].detach().clone(), sd[p + '.running_var'].detach().clone()
    # F.batch_norm is the same ATen op nn.BatchNorm2d dispatches to; its fused backward is markedly more
    # accurate in fp32 than a hand-composed mean/var/rsqrt graph when a channel's batch variance is tiny.
    y = F.batch_norm(x, rm, rv, w, b, training=train, momentum=momentum, eps=eps)
    if train and bn_updates is not None:
        bn_updates[p + '.running_mean'] = rm
        bn_updates[p + '.running_var'] = rv
        bn_updates[p + '.num_batches_tracked'] = sd[p + '.num_batches_tracked'] + 1
    return y


def expert_stem(x, sd, dom, patch_size, train, bn_updates):
    """vit.py:88-120. label stems: Upsample(4/p), strides 2,2,1,1; dense stems: Upsample(16/p), strides 2,2,2,2."""
    label = dom in LABEL_DOMAINS
    scale = (4 if label else 16) / patch_size
    if scale != 1.0:
        x = F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=True)   # nn.UpsamplingBilinear2d
    strides = (2, 2, 1, 1) if label else (2, 2, 2, 2)
    for i, s in enumerate(strides):
        x = F.conv2d(x, sd[f'conv1.{dom}.{1 + 3 * i}.weight'], None, stride=s, padding=1)
        x = torch.relu(batch_norm(x, sd, f'conv1.{dom}.{2 + 3 * i}', train, bn_updates))
    return F.conv2d(x, sd[f'conv1.{dom}.13.weight'], None)


def label_table(kind, label_info, features, background):
    """[256, 64] fp32 table: row l = the CLIP text feature post_label_process paints over label l of ONE image
    (dataset/utils.py:126-158); row 255 = BACKGROUND_FEATURES.  kind: 'seg_coco' / 'seg_ade' -> features[l] (`:128-133`,
    `:137-142`); 'obj_detection' -> features[label_info[str(l)]] (`:146-151`); 'ocr_detection' -> label_info[l]['features']
    (`:155-159`).  Labels that cannot occur in the image keep a zero row."""
    t = torch.zeros(256, background.numel(), dtype=torch.float32)
    t[255] = background.flatten().float()
    if kind in ('seg_coco', 'seg_ade'):
        n = min(255, features.shape[0])
        t[:n] = features[:n].float()
    elif kind == 'obj_detection':
        for k, v in (label_info or {}).items():
            t[int(k)] = features[v].float()
    elif kind == 'ocr_detection':
        for k, v in (label_info or {}).items():
            t[int(k)] = v['features'].flatten().float()
    else:
        raise KeyError(kind)
    return t


def post_label_process(label_map, table):
    """dataset/utils.py:126-160 for one label expert: label_map [1, H, W] integer, table [256, 64] -> [64, H, W] fp32, the
    per-pixel gather the reference writes as a loop over `unique()` labels with boolean-mask assignments"""
    return table[label_map[0].long()].permute(2, 0, 1).contiguous()


def remap_dense(x, eps=1e-6):
    """dataset/utils.py:120-121: depth / normal / edge maps are min-max remapped to [-1, 1] per sample"""
    return 2 * (x - x.min()) / (x.max() - x.min() + eps) - 1


def reference_instance_table(instance, rng):
    """vit.py:145-147: ONE random.randint(0,127) per distinct instance id over the whole batch, drawn in
    the (sorted) order of Tensor.unique(). Returns a 256-entry table label -> embedding row."""
    table = [0] * 256
    for l in instance.unique().tolist():
        table[int(l)] = rng.randint(0, 127)
    return table


def vision_encoder(sd, x, patch_size, heads, train=False, instance_table=None, bn_updates=None,
                   resampler_heads=8):
    """VisionTransformer.forward, vit.py:133-172. sd: encoder state dict (no prefix). x: expert dict in
    reference iteration order. Returns [S, B, D] (sequence first), like the reference."""
    W = sd['ln_pre.weight'].shape[0]
    pos = sd['positional_embedding']
    rgb_tok, exp_tok = None, []
    for name, val in x.items():
        dom = 'seg' if 'seg' in name else name                                   # vit.py:136
        inp = val['label'] if name == 'obj_detection' else val
        if dom == 'rgb':
            f = F.conv2d(inp, sd['conv1.rgb.weight'], None, stride=patch_size)   # vit.py:86
        else:
            f = expert_stem(inp, sd, dom, patch_size, train, bn_updates)
        if name == 'obj_detection':                                              # vit.py:141-148
            inst = F.interpolate(val['instance'].to(f.dtype), size=f.shape[2:], mode='nearest')[:, 0].long()
            tab = torch.as_tensor(instance_table, dtype=torch.long)
            f = f + sd['instance_embedding'][tab[inst]].permute(0, 3, 1, 2)
        f = f.flatten(2).transpose(1, 2)                                         # [B, g*g, D]
        if dom == 'rgb':
            rgb_tok = f + pos                                                    # vit.py:153-155
        else:
            exp_tok.append(f + interpolate_pos_embed(pos, f.shape[1]))           # vit.py:157-159
    if exp_tok:
        lat = perceiver_resampler(sd, torch.cat(exp_tok, dim=1), resampler_heads)  # vit.py:161-163
        h = torch.cat([rgb_tok, lat], dim=1)                                     # vit.py:165
    else:
        h = rgb_tok
    h = layer_norm(h, sd['ln_pre.weight'], sd['ln_pre.bias'])
    l = 0
    while f'transformer.resblocks.{l}.0.ln_1.weight' in sd:                      # vit.py:70-75
        p = f'transformer.resblocks.{l}.'
        a = layer_norm(h, sd[p + '0.ln_1.weight'], sd[p + '0.ln_1.bias'])
        h = h + packed_mha(a, a, sd[p + '0.attn.in_proj_weight'], sd[p + '0.attn.in_proj_bias'],
                           sd[p + '0.attn.out_proj.weight'], sd[p + '0.attn.out_proj.bias'], heads)
        h = adaptor(h, sd, p + '1.', norm_late=False)
        m = layer_norm(h, sd[p + '0.ln_2.weight'], sd[p + '0.ln_2.bias'])
        h = h + linear(quick_gelu(linear(m, sd, p + '0.mlp.c_fc')), sd, p + '0.mlp.c_proj')
        l += 1
    h = layer_norm(h, sd['ln_post.weight'], sd['ln_post.bias'])
    return h.transpose(0, 1)                                                     # [S, B, D]


def perceiver_resampler(sd, xf, heads=8):
    """resampler.py:33-36,46-52. xf [B, M, D] (un-normalised expert tokens, shared by all layers)."""
    B = xf.shape[0]
    lat = sd['resampler.latents'].unsqueeze(0).expand(B, -1, -1)
    l = 0
    while f'resampler.perceiver_blocks.{l}.ln_1.weight' in sd:
        p = f'resampler.perceiver_blocks.{l}.'
        q = layer_norm(lat, sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'])
        kv = torch.cat([q, layer_norm(xf, sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias'])], dim=1)
        lat = lat + packed_mha(q, kv, sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias'],
                               sd[p + 'attn.out_proj.weight'], sd[p + 'attn.out_proj.bias'], heads)
        f = layer_norm(lat, sd[p + 'ln_ff.weight'], sd[p + 'ln_ff.bias'])
        lat = lat + linear(squared_relu(linear(f, sd, p + 'mlp.c_fc')), sd, p + 'mlp.c_proj')
        l += 1
    return lat


# ------------------------------------------------------------------ decoder

def position_ids_from_input_ids(ids, pad):
    """roberta.py:38-45."""
    m = (ids != pad).to(torch.int64)
    return torch.cumsum(m, dim=1) * m + pad


def extended_attention_mask(attention_mask, dtype):
    """transformers get_extended_attention_mask with config.is_decoder=True (roberta.py:310):
    additive [B,1,T,T], 0 where (key <= query) and key not padded, finfo.min elsewhere."""
    B, T = attention_mask.shape
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    keep = causal[None, None] & attention_mask.bool()[:, None, None, :]
    return torch.zeros(B, 1, T, T, dtype=dtype).masked_fill(~keep, torch.finfo(dtype).min)


def _no_dropout(site, x):
    return x


def roberta_attention(h, kv_src, sd, p, heads, add_mask, drop=_no_dropout, site=(0, 'self')):
    """RobertaSelfAttention + RobertaSelfOutput, roberta.py:95-126,136-140.
    Training-mode dropout (roberta.py:123 on the attention probabilities, roberta.py:138 on the output projection) is applied through
    `drop(site, x)`: the caller decides the masks, so the oracle can be run with EXACTLY the masks another implementation drew
    (site = (layer, 'self_probs' | 'self_out' | 'cross_probs' | 'cross_out')).  Default: identity = eval mode."""
    B, T, H = h.shape
    dh = H // heads
    q = linear(h, sd, p + 'self.query').reshape(B, T, heads, dh).transpose(1, 2)
    k = linear(kv_src, sd, p + 'self.key').reshape(B, -1, heads, dh).transpose(1, 2)
    v = linear(kv_src, sd, p + 'self.value').reshape(B, -1, heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if add_mask is not None:
        s = torch.clamp(s + add_mask, min=torch.finfo(s.dtype).min)              # roberta.py:113-115
    pr = drop((site[0], site[1] + '_probs'), torch.softmax(s, dim=-1))           # roberta.py:123
    o = (pr @ v).transpose(1, 2).reshape(B, T, H)
    o = drop((site[0], site[1] + '_out'), linear(o, sd, p + 'output.dense'))     # roberta.py:137-138
    return layer_norm(o + h, sd[p + 'output.LayerNorm.weight'], sd[p + 'output.LayerNorm.bias'])


def roberta_mlp(h, sd, p, drop=_no_dropout, site=(0, 'mlp')):
    """RobertaIntermediate + RobertaOutput, roberta.py:160-183 (dropout on the output projection: roberta.py:180-181, site (layer, 'mlp_out'))."""
    o = drop((site[0], site[1] + '_out'), linear(gelu_erf(linear(h, sd, p + 'intermediate.dense')), sd, p + 'output.dense'))
    return layer_norm(o + h, sd[p + 'output.LayerNorm.weight'], sd[p + 'output.LayerNorm.bias'])


def text_decoder(sd, input_ids, attention_mask, enc, heads, labels=None, pad=1, label_smoothing=0.1, drop=None):
    """RobertaForCausalLMModified.forward, roberta.py:358-399.  drop=None: eval mode (dropout off).  drop=callable(site, x): training-mode
    dropout with caller-supplied masks at every nn.Dropout of the reference decoder -- ('emb',) after the embedding LayerNorm
    (roberta.py:74-75), (l, 'self_probs' / 'self_out' / 'cross_probs' / 'cross_out' / 'mlp_out') in layer l (roberta.py:123,138,181), the
    output_layer counted as layer num_hidden_layers; the Adaptor and the LM head have no dropout (utils.py:47-65, roberta.py:415-430).
    sd: decoder state dict (no prefix); enc [B,S,Dv]. Returns (logits [B,T,V], loss [B] or None)."""
    drop = drop or _no_dropout
    e = 'roberta.embeddings.'
    dtype = sd[e + 'word_embeddings.weight'].dtype
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    am = extended_attention_mask(attention_mask, dtype)
    pos_ids = position_ids_from_input_ids(input_ids, pad)
    # nn.Embedding(padding_idx=pad) (roberta.py:52,62): the pad row takes part in the forward but receives NO gradient from the
    # lookup (it matters when a pad sits inside a row, e.g. between question and answer: prismer_vqa.py:22-30)
    h = F.embedding(input_ids, sd[e + 'word_embeddings.weight'], padding_idx=pad) + sd[e + 'token_type_embeddings.weight'][0] \
        + F.embedding(pos_ids, sd[e + 'position_embeddings.weight'], padding_idx=pad)      # roberta.py:66-73
    h = drop(('emb',), layer_norm(h, sd[e + 'LayerNorm.weight'], sd[e + 'LayerNorm.bias']))     # roberta.py:74-75
    l = 0
    while f'roberta.encoder.layer.{l}.0.attention.self.query.weight' in sd:      # roberta.py:223-227
        p = f'roberta.encoder.layer.{l}.'
        h = roberta_attention(h, h, sd, p + '0.attention.', heads, am, drop, (l, 'self'))
        h = roberta_attention(h, enc, sd, p + '1.', heads, None, drop, (l, 'cross'))
        h = adaptor(h, sd, p + '2.', norm_late=True)
        h = roberta_mlp(h, sd, p + '0.', drop, (l, 'mlp'))
        l += 1
    p = 'roberta.encoder.output_layer.'                                          # roberta.py:229-231
    h = roberta_attention(h, h, sd, p + 'attention.', heads, am, drop, (l, 'self'))
    h = roberta_mlp(h, sd, p, drop, (l, 'mlp'))
    t = layer_norm(gelu_erf(linear(h, sd, 'lm_head.dense')),
                   sd['lm_head.layer_norm.weight'], sd['lm_head.layer_norm.bias'])   # roberta.py:421-425
    logits = F.linear(t, sd['lm_head.decoder.weight'], sd['lm_head.bias'])
    loss = None
    if labels is not None:
        loss = shifted_smoothed_ce(logits, labels, label_smoothing)
    return logits, loss


def shifted_smoothed_ce(logits, labels, eps=0.1):
    """roberta.py:381-387 written out: per token
    (1-eps)*nll + eps*mean_c(-log p_c), zero where label == -100, summed over tokens per sample."""
    lg = logits[:, :-1].to(torch.float64 if logits.dtype == torch.float64 else torch.float32)
    lb = labels[:, 1:]
    lp = torch.log_softmax(lg, dim=-1)
    valid = lb != -100
    nll = -lp.gather(-1, lb.clamp(min=0).unsqueeze(-1)).squeeze(-1)
    smooth = -lp.mean(dim=-1)
    tok = ((1 - eps) * nll + eps * smooth) * valid
    return tok.sum(1)


# ------------------------------------------------------------------ heads (training branches)

def caption_loss(enc_sd, dec_sd, experts, input_ids, attention_mask, labels, dims, train_bn=False,
                 instance_table=None, bn_updates=None, drop=None):
    """PrismerCaption.forward(train=True), model/prismer_caption.py:17-34, from token ids."""
    enc = vision_encoder(enc_sd, experts, dims.patch_size, dims.vit_heads, train_bn, instance_table, bn_updates)
    enc = enc.transpose(0, 1)                                                    # 'l b d -> b l d'
    logits, loss = text_decoder(dec_sd, input_ids, attention_mask, enc, dims.num_attention_heads, labels,
                                dims.pad_token_id, dims.label_smoothing, drop)
    return loss.mean(), logits, enc


def vqa_loss(enc_sd, dec_sd, experts, input_ids, attention_mask, labels, weights, dims, **kw):
    """PrismerVQA.forward(train=True), model/prismer_vqa.py:22-42: (weights * per-sample loss).mean()."""
    drop = kw.pop('drop', None)
    enc = vision_encoder(enc_sd, experts, dims.patch_size, dims.vit_heads, **kw).transpose(0, 1)
    logits, loss = text_decoder(dec_sd, input_ids, attention_mask, enc, dims.num_attention_heads, labels,
                                dims.pad_token_id, dims.label_smoothing, drop)
    return (weights * loss).mean(), logits, enc


# ------------------------------------------------------------------ training-step glue

def freeze_mask(names, mode):
    """Prismer.prepare_to_train, model/prismer.py:39-59: requires_grad by substring of the Prismer-level name."""
    out = OrderedDict()
    for n in names:
        lang = 'encoder.layer' in n and all(k not in n for k in ('1.self', '1.output', 'adaptor'))
        vis = 'transformer.resblocks' in n and 'adaptor' not in n
        if mode == 'freeze_lang':
            out[n] = not lang
        elif mode == 'freeze_vision':
            out[n] = not vis
        elif mode == 'freeze_lang_vision':
            out[n] = not (lang or vis)
        else:
            out[n] = True
    return out


def cosine_lr(it, total, init_lr, min_lr):
    """utils.py:13-17."""
    return (init_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * it / total)) + min_lr


def adamw_step(p, g, m, v, step, lr, wd=0.05, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor math (train_caption.py:111-112): decoupled weight decay."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mh = m / (1 - b1 ** step)
    vh = v / (1 - b2 ** step)
    return p - lr * mh / (vh.sqrt() + eps), m, v


def rank_answers(dec_sd, enc_out, start_ids, start_att, ans_ids, ans_att, k_test, heads, pad=1):
    """`inference='rank'` of the caption and VQA heads (model/prismer_caption.py:59-112, model/prismer_vqa.py:64-113):
    (1) logits of the prompt's last position -> softmax -> probability of every candidate answer's FIRST token -> top-k answers;
    (2) prompt (tiled k times, reference `tile`: item-major) || each of the k answers, targets = the answer span only (pad -> -100);
    (3) length-normalised log-probability  -loss / #target tokens  (loss = per-sample SUM of the shifted, label-smoothed CE of
        text_decoder, roberta.py:381-387) -> arg-max over k -> the answer index.
    enc_out: [B, S, D] encoder states (batch first).  Returns (best answer index [B], topk_ids [B, k], log-probs [B, k])."""
    B = start_ids.shape[0]
    logits, _ = text_decoder(dec_sd, start_ids, start_att, enc_out, heads)
    prob_first = torch.softmax(logits[:, -1, :].float(), dim=1).index_select(1, ans_ids[:, 0])
    _, topk_ids = prob_first.topk(k_test, dim=1)
    a_ids = torch.cat([ans_ids.index_select(0, t) for t in topk_ids], 0)
    a_att = torch.cat([ans_att.index_select(0, t) for t in topk_ids], 0)
    rep = torch.arange(B).repeat_interleave(k_test)                     # reference tile(): rows b*k .. b*k+k-1 are item b
    input_ids = torch.cat([start_ids[rep], a_ids], 1).long()
    att = torch.cat([start_att[rep], a_att], 1)
    targets = input_ids.masked_fill(input_ids == pad, -100)
    targets[:, :-ans_ids.shape[1]] = -100
    _, loss = text_decoder(dec_sd, input_ids, att, enc_out[rep], heads, targets, pad=pad)
    lp = (-loss / (targets != -100).sum(-1)).view(-1, k_test)
    best = lp.argmax(1)
    return topk_ids[torch.arange(B), best], topk_ids, lp
