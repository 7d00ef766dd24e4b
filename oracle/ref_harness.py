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
