"""Harness that instantiates the REAL reference module classes from /root/reference  --  TEST INFRASTRUCTURE.

Only usable in the build container (the GPU box has no /root/reference).  Used by
tests/golden/make_golden.py to mint the golden fixtures and by tests/test_oracle_golden.py to check the
oracle restatement against the live reference.  Follows SURVEY App. D: the factories
(load_encoder / load_decoder, vit.py:175, roberta.py:433) need the network, so the classes are built
directly and filled with prismer_amd.synth weights; the LM head is tied explicitly (transformers-4.26
semantics, roberta.py:352-353).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('PRISMER_REFERENCE_ROOT', '/root/reference')


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'model', 'modules', 'vit.py'))


def _import_reference():
    if 'clip' not in sys.modules:                     # vit.py:10 imports clip.clip._download only
        clip = types.ModuleType('clip'); cc = types.ModuleType('clip.clip')
        cc._download = lambda *a, **k: None
        clip.clip = cc
        sys.modules['clip'] = clip; sys.modules['clip.clip'] = cc
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import model.modules.vit as V
    import model.modules.roberta as R
    return V, R


def build_reference(dims, enc_sd, dec_sd):
    """Returns (VisionTransformer, RobertaForCausalLMModified) in eval mode with the given weights."""
    import torch
    from transformers import RobertaConfig
    V, R = _import_reference()
    enc = V.VisionTransformer(dims.image_resolution, dims.patch_size, dims.width, dims.vit_layers,
                              dims.vit_heads, dict(dims.experts))
    dec = R.RobertaForCausalLMModified(RobertaConfig.from_dict(dims.roberta_config_dict()))
    dec.lm_head.decoder.weight = dec.roberta.embeddings.word_embeddings.weight
    missing, unexpected = enc.load_state_dict(enc_sd, strict=True), None
    sd = {k: v for k, v in dec_sd.items()}
    res = dec.load_state_dict(sd, strict=False)
    bad = [k for k in res.missing_keys if 'token_type_ids' not in k]
    assert not bad and not res.unexpected_keys, (bad, res.unexpected_keys)
    assert dec.lm_head.decoder.weight.data_ptr() == dec.roberta.embeddings.word_embeddings.weight.data_ptr()
    assert dec.lm_head.decoder.bias is dec.lm_head.bias
    enc.eval(); dec.eval()
    return enc, dec


def reference_freeze(enc, dec, mode):
    """Runs the reference's own Prismer.prepare_to_train (model/prismer.py:39-59) on a holder module that
    carries the two sub-modules under the Prismer-level attribute names. Returns the holder."""
    import torch.nn as nn
    _import_reference()
    from model.prismer import Prismer

    class Holder(nn.Module):
        pass
    h = Holder()
    h.expert_encoder = enc
    h.text_decoder = dec
    Prismer.prepare_to_train(h, mode)
    return h


def dropout_sites(dec):
    """{module name: oracle dropout site} for EVERY nn.Dropout of the reference decoder (roberta.py:57,93,134,177): the site naming of
    oracle.prismer_oracle.text_decoder.  Raises when the reference has a Dropout this mapping does not know."""
    import re
    import torch.nn as nn
    nl = dec.config.num_hidden_layers
    kinds = {'0.attention.self': 'self_probs', '0.attention.output': 'self_out', '1.self': 'cross_probs', '1.output': 'cross_out',
             '0.output': 'mlp_out'}
    final = {'attention.self': 'self_probs', 'attention.output': 'self_out', 'output': 'mlp_out'}
    sites = {}
    for name, mod in dec.named_modules():
        if not isinstance(mod, nn.Dropout):
            continue
        m = re.fullmatch(r'roberta\.encoder\.layer\.(\d+)\.(.+)\.dropout', name)
        f = re.fullmatch(r'roberta\.encoder\.output_layer\.(.+)\.dropout', name)
        if name == 'roberta.embeddings.dropout':
            sites[name] = ('emb',)
        elif m and m.group(2) in kinds:
            sites[name] = (int(m.group(1)), kinds[m.group(2)])
        elif f and f.group(1) in final:
            sites[name] = (nl, final[f.group(1)])
        else:
            raise KeyError(f'reference decoder has an nn.Dropout the oracle does not model: {name}')
    return sites


def patch_dropout(dec, drop):
    """Replaces the forward of every nn.Dropout of the reference decoder by `drop(site, x)` (training-mode dropout with the caller's
    masks) and puts the decoder in train() mode.  Returns the site map."""
    sites = dropout_sites(dec)
    mods = dict(dec.named_modules())
    for name, site in sites.items():
        mods[name].forward = (lambda x, _s=site: drop(_s, x))
    dec.train()
    return sites
