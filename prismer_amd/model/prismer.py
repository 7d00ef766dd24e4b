"""Prismer base model (reference: model/prismer.py:15-94): builds tokenizer + HIP-backed encoder/decoder with the
reference attribute names (`expert_encoder`, `text_decoder`, `tokenizer`, `ignored_modules`), applies the
name-substring freeze policy and exposes the FSDP ignore list."""
import json
import os

import torch.nn as nn

from ..config import expert_channels
from ..modules.roberta import load_decoder
from ..modules.vit import load_encoder

# configs/prismer.json of the reference (model dims; read at model/prismer.py:29). Kept in code so that the package
# does not depend on the caller's working directory; a JSON file at `configs/prismer.json` overrides it.
_ROBERTA_COMMON = dict(attention_probs_dropout_prob=0.1, bos_token_id=0, eos_token_id=2, hidden_act='gelu', hidden_dropout_prob=0.1,
                       initializer_range=0.02, layer_norm_eps=1e-05, max_position_embeddings=514, pad_token_id=1, type_vocab_size=1,
                       vocab_size=50265, num_decoder_layers=4, is_decoder=True)
PRISMER_CONFIGS = {
    'prismer_base': {'roberta_model': dict(_ROBERTA_COMMON, hidden_size=768, vision_hidden_size=768, intermediate_size=3072,
                                            model_name='roberta-base', num_attention_heads=12, num_hidden_layers=12),
                     'vit_model': 'ViT-B/16'},
    'prismer_large': {'roberta_model': dict(_ROBERTA_COMMON, hidden_size=1024, vision_hidden_size=1024, intermediate_size=4096,
                                             model_name='roberta-large', num_attention_heads=16, num_hidden_layers=24),
                      'vit_model': 'ViT-L/14@336px'},
    'prismer_huge': {'roberta_model': dict(_ROBERTA_COMMON, hidden_size=1024, vision_hidden_size=1280, intermediate_size=4096,
                                            model_name='roberta-large', num_attention_heads=16, num_hidden_layers=24),
                     'vit_model': 'ViT-H/14'},
}


class _Cfg:
    def __init__(self, d):
        self.__dict__.update(d)


def _load_tokenizer(name):
    """RobertaTokenizer.from_pretrained(name) of model/prismer.py:32, offline only. Returns None when the vocabulary
    files are not on disk (callers must then pass token ids instead of strings)."""
    try:
        from transformers import RobertaTokenizer
        tok = RobertaTokenizer.from_pretrained(name, local_files_only=True)
        if tok.pad_token_id != 1 or len(tok) < 50000:      # offline stub tokenizer: unusable
            return None
        return tok
    except Exception:
        return None


class Prismer(nn.Module):
    def __init__(self, config):
        super().__init__()
        experts = config['experts'] if config.get('experts', 'none') != 'none' else []
        self.experts = dict(expert_channels(experts))
        table = PRISMER_CONFIGS
        if os.path.isfile('configs/prismer.json'):
            table = json.load(open('configs/prismer.json', 'r'))
        prismer_config = table[config['prismer_model']]
        roberta_config = _Cfg(prismer_config['roberta_model'])
        self.tokenizer = _load_tokenizer(prismer_config['roberta_model']['model_name'])
        self.expert_encoder = load_encoder(prismer_config['vit_model'], experts=self.experts, image_resolution=config['image_resolution'],
                                           checkpoint_path=config.get('clip_checkpoint'))
        self.text_decoder = load_decoder(prismer_config['roberta_model']['model_name'], config=roberta_config,
                                         checkpoint_path=config.get('roberta_checkpoint'))
        self.prepare_to_train(config.get('freeze', 'none'))
        self.ignored_modules = self.get_ignored_modules(config.get('freeze', 'none'))

    def prepare_to_train(self, mode='none'):
        """requires_grad by substring of the parameter NAME (model/prismer.py:39-59)."""
        for name, params in self.named_parameters():
            lang = 'encoder.layer' in name and all(key not in name for key in ['1.self', '1.output', 'adaptor'])
            vis = 'transformer.resblocks' in name and 'adaptor' not in name
            if mode == 'freeze_lang':
                params.requires_grad = not lang
            elif mode == 'freeze_vision':
                params.requires_grad = not vis
            elif mode == 'freeze_lang_vision':
                params.requires_grad = not (lang or vis)
            else:
                params.requires_grad = True

    def get_ignored_modules(self, mode='none'):
        """FSDP ignored_modules list (model/prismer.py:61-94)."""
        lang, vis = [], []
        for lyr in self.text_decoder.roberta.encoder.layer:
            lang += [lyr[0].attention, lyr[0].intermediate, lyr[0].output]
        for blk in self.expert_encoder.transformer.resblocks:
            vis += [blk[0].attn, blk[0].mlp, blk[0].ln_1, blk[0].ln_2]
        return {'freeze_lang': lang, 'freeze_vision': vis, 'freeze_lang_vision': lang + vis}.get(mode)

    # ---- tokenisation helpers shared by the heads --------------------------------------------------
    def _tokenize(self, text, **kw):
        if self.tokenizer is None:
            raise RuntimeError('no RoBERTa vocabulary on disk (offline): pass token ids (dict with input_ids / attention_mask) '
                               'instead of strings')
        return self.tokenizer(text, **kw)
