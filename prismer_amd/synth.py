"""Deterministic synthetic weights and inputs for the Prismer hot path.

There is no network in the build or bench environment, so neither the CLIP / RoBERTa checkpoints
(reference: model/modules/vit.py:179-184, roberta.py:436) nor the tokenizer are reachable.  Everything
here is generated from a counter-based integer hash (splitmix64) so that the SAME tensors are produced
on any machine with no dependence on torch / libm RNG streams: the golden fixtures under tests/golden/
store only the reference's OUTPUTS; weights and inputs are regenerated from (name, seed).

State-dict key names / shapes follow the reference contract (model/modules/vit.py:78-131,
resampler.py:15-44, roberta.py:48-240,409-419, utils.py:48-58).
"""
from collections import OrderedDict
import zlib

import numpy as np
import torch

from .config import PrismerDims, LABEL_DOMAINS

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _key(name: str, seed: int) -> np.uint64:
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    return np.uint64(((seed & 0xFFFFFFFF) << 32) | h)


def uniform_pm1(name: str, shape, seed: int = 0) -> torch.Tensor:
    """U(-1, 1) float32, exactly reproducible (24 random mantissa bits / 2^23 - 1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        bits = _splitmix64(ctr * np.uint64(0xD1342543DE82EF95) + _key(name, seed))
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0   # 24 bits -> [-1, 1)
    return torch.from_numpy(u.astype(np.float32).reshape(shape))


def randint(name: str, shape, lo: int, hi: int, seed: int = 0) -> torch.Tensor:
    """Integers in [lo, hi) int64."""
    n = int(np.prod(shape)) if len(shape) else 1
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        bits = _splitmix64(ctr * np.uint64(0xD1342543DE82EF95) + _key(name, seed))
    v = (bits >> np.uint64(11)) % np.uint64(hi - lo)
    return torch.from_numpy(v.astype(np.int64).reshape(shape) + lo)


def uniform_std(name, shape, std, seed=0):
    return uniform_pm1(name, shape, seed) * (std * 3.0 ** 0.5)


# ----------------------------------------------------------------------------------------------
# state-dict specification
# ----------------------------------------------------------------------------------------------

def encoder_spec(d: PrismerDims) -> "OrderedDict[str, tuple]":
    """key -> (shape, kind) for VisionTransformer (vit.py:78-131)."""
    W = d.width
    s = OrderedDict()
    s['positional_embedding'] = ((d.num_rgb_tokens, W), 'pos')
    if 'obj_detection' in d.experts:
        s['instance_embedding'] = ((128, W), 'pos')
    for e, cin in d.experts.items():
        if e == 'rgb':
            s['conv1.rgb.weight'] = ((W, cin, d.patch_size, d.patch_size), 'conv')
            continue
        chans = [cin if e not in LABEL_DOMAINS else 64, W // 8, W // 4, W // 2, W]
        for i in range(4):
            s[f'conv1.{e}.{1 + 3 * i}.weight'] = ((chans[i + 1], chans[i], 3, 3), 'conv')
            bn = f'conv1.{e}.{2 + 3 * i}'
            s[bn + '.weight'] = ((chans[i + 1],), 'ln_w')
            s[bn + '.bias'] = ((chans[i + 1],), 'ln_b')
            s[bn + '.running_mean'] = ((chans[i + 1],), 'bn_rm')
            s[bn + '.running_var'] = ((chans[i + 1],), 'bn_rv')
            s[bn + '.num_batches_tracked'] = ((), 'i64zero')
        s[f'conv1.{e}.13.weight'] = ((W, W, 1, 1), 'conv')
    for l in range(d.vit_layers):
        p = f'transformer.resblocks.{l}.'
        s[p + '0.attn.in_proj_weight'] = ((3 * W, W), 'linear')
        s[p + '0.attn.in_proj_bias'] = ((3 * W,), 'bias')
        s[p + '0.attn.out_proj.weight'] = ((W, W), 'linear')
        s[p + '0.attn.out_proj.bias'] = ((W,), 'bias')
        s[p + '0.mlp.c_fc.weight'] = ((4 * W, W), 'linear')
        s[p + '0.mlp.c_fc.bias'] = ((4 * W,), 'bias')
        s[p + '0.mlp.c_proj.weight'] = ((W, 4 * W), 'linear')
        s[p + '0.mlp.c_proj.bias'] = ((W,), 'bias')
        for ln in ('ln_1', 'ln_2'):
            s[p + f'0.{ln}.weight'] = ((W,), 'ln_w')
            s[p + f'0.{ln}.bias'] = ((W,), 'ln_b')
        for proj in ('down_proj', 'up_proj'):
            s[p + f'1.adaptor.{proj}.weight'] = ((W, W), 'linear')
            s[p + f'1.adaptor.{proj}.bias'] = ((W,), 'bias')
        s[p + '1.adaptor_ln.weight'] = ((W,), 'ln_w')
        s[p + '1.adaptor_ln.bias'] = ((W,), 'ln_b')
    if d.has_experts:
        s['resampler.latents'] = ((d.num_latents, W), 'pos')
        for l in range(d.resampler_layers):
            p = f'resampler.perceiver_blocks.{l}.'
            s[p + 'attn.in_proj_weight'] = ((3 * W, W), 'linear')
            s[p + 'attn.in_proj_bias'] = ((3 * W,), 'bias')
            s[p + 'attn.out_proj.weight'] = ((W, W), 'linear')
            s[p + 'attn.out_proj.bias'] = ((W,), 'bias')
            s[p + 'mlp.c_fc.weight'] = ((4 * W, W), 'linear')
            s[p + 'mlp.c_fc.bias'] = ((4 * W,), 'bias')
            s[p + 'mlp.c_proj.weight'] = ((W, 4 * W), 'linear')
            s[p + 'mlp.c_proj.bias'] = ((W,), 'bias')
            for ln in ('ln_1', 'ln_2', 'ln_ff'):
                s[p + f'{ln}.weight'] = ((W,), 'ln_w')
                s[p + f'{ln}.bias'] = ((W,), 'ln_b')
    for ln in ('ln_pre', 'ln_post'):
        s[f'{ln}.weight'] = ((W,), 'ln_w')
        s[f'{ln}.bias'] = ((W,), 'ln_b')
    return s


def decoder_spec(d: PrismerDims) -> "OrderedDict[str, tuple]":
    """key -> (shape, kind) for RobertaForCausalLMModified (roberta.py:48-76,201-210,336-419)."""
    H, Hv, I, V = d.hidden_size, d.vision_hidden_size, d.intermediate_size, d.vocab_size
    s = OrderedDict()
    e = 'roberta.embeddings.'
    s[e + 'position_ids'] = ((1, d.max_position_embeddings), 'arange')
    s[e + 'word_embeddings.weight'] = ((V, H), 'emb')
    s[e + 'position_embeddings.weight'] = ((d.max_position_embeddings, H), 'emb')
    s[e + 'token_type_embeddings.weight'] = ((d.type_vocab_size, H), 'emb')
    s[e + 'LayerNorm.weight'] = ((H,), 'ln_w')
    s[e + 'LayerNorm.bias'] = ((H,), 'ln_b')

    def attn(p, kv_in):
        s[p + 'self.query.weight'] = ((H, H), 'linear'); s[p + 'self.query.bias'] = ((H,), 'bias')
        s[p + 'self.key.weight'] = ((H, kv_in), 'linear'); s[p + 'self.key.bias'] = ((H,), 'bias')
        s[p + 'self.value.weight'] = ((H, kv_in), 'linear'); s[p + 'self.value.bias'] = ((H,), 'bias')
        s[p + 'output.dense.weight'] = ((H, H), 'linear'); s[p + 'output.dense.bias'] = ((H,), 'bias')
        s[p + 'output.LayerNorm.weight'] = ((H,), 'ln_w'); s[p + 'output.LayerNorm.bias'] = ((H,), 'ln_b')

    def layer(p):
        attn(p + 'attention.', H)
        s[p + 'intermediate.dense.weight'] = ((I, H), 'linear'); s[p + 'intermediate.dense.bias'] = ((I,), 'bias')
        s[p + 'output.dense.weight'] = ((H, I), 'linear'); s[p + 'output.dense.bias'] = ((H,), 'bias')
        s[p + 'output.LayerNorm.weight'] = ((H,), 'ln_w'); s[p + 'output.LayerNorm.bias'] = ((H,), 'ln_b')

    for l in range(d.num_hidden_layers):
        p = f'roberta.encoder.layer.{l}.'
        layer(p + '0.')
        attn(p + '1.', Hv)
        for proj in ('down_proj', 'up_proj'):
            s[p + f'2.adaptor.{proj}.weight'] = ((H, H), 'linear')
            s[p + f'2.adaptor.{proj}.bias'] = ((H,), 'bias')
        s[p + '2.adaptor_ln.weight'] = ((H,), 'ln_w'); s[p + '2.adaptor_ln.bias'] = ((H,), 'ln_b')
    layer('roberta.encoder.output_layer.')
    s['lm_head.bias'] = ((V,), 'bias')
    s['lm_head.dense.weight'] = ((H, H), 'linear'); s['lm_head.dense.bias'] = ((H,), 'bias')
    s['lm_head.layer_norm.weight'] = ((H,), 'ln_w'); s['lm_head.layer_norm.bias'] = ((H,), 'ln_b')
    # lm_head.decoder.weight is TIED to word_embeddings.weight and lm_head.decoder.bias IS lm_head.bias
    # (transformers 4.26 semantics, roberta.py:352-353,417-419); they are aliases, not independent tensors.
    s['lm_head.decoder.weight'] = ((V, H), 'alias:roberta.embeddings.word_embeddings.weight')
    s['lm_head.decoder.bias'] = ((V,), 'alias:lm_head.bias')
    return s


def _make(name, shape, kind, seed):
    if kind == 'linear':
        fan_in = shape[1]
        return uniform_std(name, shape, 0.7 / fan_in ** 0.5, seed)
    if kind == 'conv':
        fan_in = shape[1] * shape[2] * shape[3]
        return uniform_std(name, shape, (2.0 / fan_in) ** 0.5, seed)
    if kind == 'bias':
        return uniform_std(name, shape, 0.02, seed)
    if kind == 'ln_w':
        return 1.0 + uniform_std(name, shape, 0.1, seed)
    if kind == 'ln_b':
        return uniform_std(name, shape, 0.05, seed)
    if kind == 'emb':
        return uniform_std(name, shape, 0.05, seed)
    if kind == 'pos':
        return uniform_std(name, shape, shape[-1] ** -0.5, seed)
    if kind == 'bn_rm':
        return uniform_std(name, shape, 0.1, seed)
    if kind == 'bn_rv':
        return 1.0 + 0.5 * uniform_pm1(name, shape, seed)
    if kind == 'i64zero':
        return torch.zeros(shape, dtype=torch.int64)
    if kind == 'arange':
        return torch.arange(shape[-1], dtype=torch.int64).reshape(shape)
    raise ValueError(kind)


def synth_state_dict(spec, seed: int = 0, prefix: str = '') -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, (shape, kind) in spec.items():
        if kind.startswith('alias:'):
            out[prefix + k] = out[prefix + kind[6:]]
        else:
            out[prefix + k] = _make(k, shape, kind, seed)
    return out


def synth_encoder_state(d: PrismerDims, seed=0):
    return synth_state_dict(encoder_spec(d), seed)


def synth_decoder_state(d: PrismerDims, seed=0):
    sd = synth_state_dict(decoder_spec(d), seed)
    # padding rows are zero in the reference's Embedding(padding_idx=...) init (roberta.py:51,63)
    sd['roberta.embeddings.word_embeddings.weight'][d.pad_token_id].zero_()
    sd['roberta.embeddings.position_embeddings.weight'][d.pad_token_id].zero_()
    return sd


# ----------------------------------------------------------------------------------------------
# synthetic inputs (SURVEY 8d "synthetic inputs"; input contract dataset/utils.py:30-71,117-160)
# ----------------------------------------------------------------------------------------------

def synth_experts(d: PrismerDims, batch: int, seed: int = 1234, expert_names=None) -> "OrderedDict":
    """Expert dict in the reference iteration order: 'rgb' first then config['experts'] order
    (dataset/utils.py:69,81). Label experts are piecewise-constant 64-channel maps: a uint8 label map of
    random rectangles over background 255, gathered through a [256,64] table (std 0.75)."""
    R, E = d.image_resolution, d.expert_resolution
    x = OrderedDict()
    x['rgb'] = uniform_std('in.rgb', (batch, 3, R, R), 1.0, seed)
    if expert_names is None:
        expert_names = [k for k in d.experts if k != 'rgb']
        expert_names = ['seg_coco' if k == 'seg' else k for k in expert_names]
    for name in expert_names:
        dom = 'seg' if 'seg' in name else name
        if dom in ('depth', 'edge'):
            x[name] = uniform_pm1('in.' + name, (batch, 1, E, E), seed)
        elif dom == 'normal':
            x[name] = uniform_pm1('in.' + name, (batch, 3, E, E), seed)
        else:
            lab = torch.full((batch, E, E), 255, dtype=torch.int64)
            rect = randint('in.rect.' + name, (batch, 8, 5), 0, 1 << 20, seed)
            for b in range(batch):
                for r in range(8):
                    y0 = int(rect[b, r, 0]) % E; x0 = int(rect[b, r, 1]) % E
                    h = 1 + int(rect[b, r, 2]) % max(1, E // 2); w = 1 + int(rect[b, r, 3]) % max(1, E // 2)
                    lab[b, y0:y0 + h, x0:x0 + w] = int(rect[b, r, 4]) % 200
            table = uniform_std('in.table.' + name, (256, 64), 0.75, seed)
            m = table[lab].permute(0, 3, 1, 2).contiguous()            # [B,64,E,E]
            if dom == 'obj_detection':
                x[name] = {'label': m, 'instance': lab.unsqueeze(1).clone()}   # dataset/utils.py:149
            else:
                x[name] = m
    return x


def synth_text(d: PrismerDims, batch: int, T: int, seed: int = 1234, ragged: bool = False, prompt_length: int = 4):
    """input_ids / attention_mask / labels as PrismerCaption.forward builds them
    (model/prismer_caption.py:21-26): <s>=0 ... </s>=2, pad=1, labels: pad -> -100, prompt -> -100."""
    ids = randint('in.ids', (batch, T), 3, d.vocab_size, seed)
    ids[:, 0] = d.bos_token_id
    ids[:, 1:prompt_length] = torch.tensor([83, 2170, 9][:max(0, prompt_length - 1)]) % d.vocab_size  # "A picture of"-like fixed ids
    mask = torch.ones(batch, T, dtype=torch.int64)
    lens = [T] * batch
    if ragged:
        lo = min(T, prompt_length + 3)
        lens = [T - ((3 * b + 2) % (T - lo + 1)) if b else T for b in range(batch)]   # padding='longest': row 0 is full
    for b, L in enumerate(lens):
        ids[b, L - 1] = d.eos_token_id
        ids[b, L:] = d.pad_token_id
        mask[b, L:] = 0
    labels = ids.masked_fill(ids == d.pad_token_id, -100)
    labels[:, :prompt_length] = -100
    return ids, mask, labels
