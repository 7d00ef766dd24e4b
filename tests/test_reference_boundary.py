"""CPU, build container only (needs /root/reference): the drop-in boundary checked on the REFERENCE's own code.

  * model/prismer.py of the reference, imported unchanged, with ONLY the two factory imports (`load_encoder`, `load_decoder`,
    model/prismer.py:10-11,33-34) pointed at prismer_amd -- INTEGRATION.md section 1 as a test: construction, the substring
    freeze rule, `get_ignored_modules`, `len(model.expert_encoder.positional_embedding)` (train_caption.py:98-99), strict
    loading of a state dict produced by the reference's own module classes (train_caption.py:100,108).
  * checkpoint surgery: the reference's load_encoder (vit.py:175-225) run on a synthetic CLIP-shaped TorchScript
    archive and its load_decoder (roberta.py:433-452) run on a RobertaForMaskedLM, against prismer_amd's `checkpoint_path=` path:
    same keys, same tensors, same bicubic positional re-grid."""
import json
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

from oracle import ref_harness as RH

pytestmark = pytest.mark.skipif(not RH.available(), reason='reference not mounted (GPU box)')

TINY_ROBERTA = dict(attention_probs_dropout_prob=0.1, bos_token_id=0, eos_token_id=2, hidden_act='gelu', hidden_dropout_prob=0.1,
                    hidden_size=128, vision_hidden_size=128, initializer_range=0.02, intermediate_size=256, layer_norm_eps=1e-5,
                    max_position_embeddings=64, model_name='roberta-base', num_attention_heads=2, num_hidden_layers=2, pad_token_id=1,
                    type_vocab_size=1, vocab_size=211, is_decoder=True)


class _Attn(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * w, w) * 0.02)
        self.in_proj_bias = nn.Parameter(torch.randn(3 * w) * 0.02)
        self.out_proj = nn.Linear(w, w)

    def forward(self, x):
        return x


class _Mlp(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.c_fc, self.c_proj = nn.Linear(w, 4 * w), nn.Linear(4 * w, w)

    def forward(self, x):
        return x


class _Block(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.attn, self.ln_1, self.mlp, self.ln_2 = _Attn(w), nn.LayerNorm(w), _Mlp(w), nn.LayerNorm(w)

    def forward(self, x):
        return x


class _Transformer(nn.Module):
    def __init__(self, w, layers):
        super().__init__()
        self.resblocks = nn.Sequential(*[_Block(w) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class _Visual(nn.Module):
    """the `visual.*` part of an OpenAI CLIP archive (names and shapes), tiny: width 128, patch 16, 7x7 grid (+ class token)"""

    def __init__(self, w=128, layers=2, patch=16, grid=7):
        super().__init__()
        self.conv1 = nn.Conv2d(3, w, patch, patch, bias=False)
        self.class_embedding = nn.Parameter(torch.randn(w) * 0.02)
        self.positional_embedding = nn.Parameter(torch.randn(grid * grid + 1, w) * 0.02)
        self.ln_pre, self.transformer, self.ln_post = nn.LayerNorm(w), _Transformer(w, layers), nn.LayerNorm(w)
        self.proj = nn.Parameter(torch.randn(w, 64) * 0.02)

    def forward(self, x):
        return x


class _Clip(nn.Module):
    def __init__(self):
        super().__init__()
        self.visual = _Visual()
        self.token_embedding = nn.Embedding(10, 8)          # non-visual entries the surgery must drop
        self.logit_scale = nn.Parameter(torch.ones([]))

    def forward(self, x):
        return x


@pytest.fixture(scope='module')
def ref():
    V, R = RH._import_reference()
    return V, R


def test_checkpoint_surgery_matches_reference_load_encoder(ref, tmp_path):
    V, _ = ref
    import prismer_amd.modules.vit as MV
    torch.manual_seed(0)
    clip_model = _Clip()
    arch = str(tmp_path / 'clip.pt')
    torch.jit.script(clip_model).save(arch)
    experts = {'rgb': 3, 'depth': 1, 'seg': 64, 'obj_detection': 64}
    old = V._download
    V._download = lambda url, root=None: arch                    # vit.py:179: the only network access
    try:
        torch.manual_seed(1)
        r = V.load_encoder('ViT-B/16', experts=dict(experts), image_resolution=224)   # 7x7 -> 14x14 bicubic re-grid on load
    finally:
        V._download = old
    sd_path = str(tmp_path / 'clip_sd.pt')
    torch.save(clip_model.state_dict(), sd_path)
    torch.manual_seed(1)
    m = MV.load_encoder('ViT-B/16', experts=dict(experts), image_resolution=224, checkpoint_path=sd_path)
    rsd, msd = r.state_dict(), m.state_dict()
    assert list(rsd) == list(msd)                                 # App. E key contract incl. order
    conv = MV.convert_clip_state_dict(clip_model.state_dict())
    loaded = [k for k in conv if k in rsd]
    assert len(loaded) == 1 + 1 + 4 + 2 * 12                      # conv1.rgb, pos-embed, ln_pre/post, 2 blocks x 12 tensors
    for k in loaded:
        assert torch.equal(rsd[k], msd[k]), k
    assert msd['positional_embedding'].shape == (196, 128)
    assert m.width == r.conv1['rgb'].weight.shape[0] == 128 and m.layers == 2 and m.heads == 2 and m.patch_size == 16
    assert 'proj' not in msd and 'class_embedding' not in msd


def test_checkpoint_surgery_matches_reference_load_decoder(ref, tmp_path):
    _, R = ref
    import prismer_amd.modules.roberta as MR
    from transformers import RobertaConfig, RobertaForMaskedLM
    cfg = RobertaConfig.from_dict(TINY_ROBERTA)
    torch.manual_seed(0)
    mlm_cfg = RobertaConfig.from_dict(dict(TINY_ROBERTA, is_decoder=False))
    mlm = RobertaForMaskedLM(mlm_cfg)
    old = R.RobertaForMaskedLM.from_pretrained
    R.RobertaForMaskedLM.from_pretrained = classmethod(lambda cls, name, cache_dir=None: mlm)     # roberta.py:436
    try:
        r = R.load_decoder('roberta-base', cfg)
    finally:
        R.RobertaForMaskedLM.from_pretrained = old
    path = str(tmp_path / 'mlm.pt')
    torch.save(mlm.state_dict(), path)
    m = MR.load_decoder('roberta-base', cfg, checkpoint_path=path)
    rsd, msd = r.state_dict(), m.state_dict()
    conv = MR.convert_roberta_state_dict(mlm.state_dict())
    common = [k for k in conv if k in rsd and k in msd]
    assert len(common) >= 5 + 2 * 16 + 5                           # embeddings, 2 layers x 16 tensors, LM head
    for k in common:
        assert torch.equal(rsd[k], msd[k]), k
    assert set(k for k in rsd if 'token_type_ids' not in k) == set(k for k in msd if 'token_type_ids' not in k)
    # never initialised from RoBERTa (roberta.py:210,440-447): cross-attention, adaptors, output_layer
    assert not any(('.1.self.' in k or 'adaptor' in k or 'output_layer' in k) for k in conv)


def test_reference_prismer_class_runs_on_prismer_amd_factories(ref, monkeypatch):
    """INTEGRATION.md section 1, executed: the reference's Prismer class with the two factory names rebound."""
    import prismer_amd.modules.roberta as MR
    import prismer_amd.modules.vit as MV
    monkeypatch.chdir(RH.REFERENCE_ROOT)                             # model/prismer.py reads configs/prismer.json from the cwd
    import model.prismer as RP
    import transformers
    small = {'prismer_small': {'roberta_model': TINY_ROBERTA, 'vit_model': 'ViT-B/16'}}
    real_load = json.load
    monkeypatch.setattr(RP.json, 'load', lambda f: small)
    monkeypatch.setattr(RP.RobertaTokenizer, 'from_pretrained', classmethod(lambda cls, name: 'tokenizer-stub'))
    cfgd = {'experts': ['depth', 'normal', 'seg_coco', 'edge', 'obj_detection', 'ocr_detection'], 'prismer_model': 'prismer_small',
            'image_resolution': 224, 'freeze': 'freeze_vision'}

    def small_vit(mod):
        def f(name, experts, image_resolution):
            return mod.VisionTransformer(image_resolution, 16, 128, 2, 2, experts)
        return f
    # (a) the reference's own classes (factories bypassed: no network) -> the state dict a reference checkpoint holds
    monkeypatch.setattr(RP, 'load_encoder', small_vit(ref[0]))
    monkeypatch.setattr(RP, 'load_decoder', lambda name, config: ref[1].RobertaForCausalLMModified(config))
    torch.manual_seed(0)
    ref_model = RP.Prismer(cfgd)
    # transformers-4.26 semantics (SURVEY 8c trap 1): the LM-head weight IS the word-embedding table; 5.x leaves them untied
    ref_model.text_decoder.lm_head.decoder.weight = ref_model.text_decoder.roberta.embeddings.word_embeddings.weight
    ref_sd = ref_model.state_dict()
    # (b) the SAME class with prismer_amd's factories
    monkeypatch.setattr(RP, 'load_encoder', small_vit(MV))
    monkeypatch.setattr(RP, 'load_decoder', lambda name, config: MR.load_decoder(name, config))
    ours = RP.Prismer(cfgd)
    assert type(ours.expert_encoder).__module__.startswith('prismer_amd') and type(ours.text_decoder).__module__.startswith('prismer_amd')
    res = ours.load_state_dict(ref_sd, strict=True)                  # train_caption.py:100,108
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in ours.state_dict().items():
        assert torch.equal(v, ref_sd[k]), k
    assert len(ours.expert_encoder.positional_embedding) == len(ref_model.expert_encoder.positional_embedding) == 196   # train_caption.py:98-99
    a = {n: p.requires_grad for n, p in ours.named_parameters()}
    b = {n: p.requires_grad for n, p in ref_model.named_parameters()}
    assert a == b and not all(a.values()) and any(a.values())        # prepare_to_train ran on OUR parameter names
    assert len(ours.ignored_modules) == len(ref_model.ignored_modules) == 2 * 4
    assert [type(m).__name__ for m in ours.ignored_modules] == [type(m).__name__ for m in ref_model.ignored_modules]
    # FSDP wrap classes importable from the same module paths (train_caption.py:71-73)
    from prismer_amd.modules.resampler import PerceiverAttentionBlock     # noqa: F401
    from prismer_amd.modules.roberta import RobertaLayer                  # noqa: F401
    from prismer_amd.modules.vit import ResidualAttentionBlock            # noqa: F401
    assert hasattr(ours.text_decoder, 'generate')                        # prismer_caption.py:45
