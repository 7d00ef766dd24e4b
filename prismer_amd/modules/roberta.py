"""Language decoder shells (reference: model/modules/roberta.py).

Class names, constructor arguments, attribute paths and state-dict keys follow the reference
(RobertaEmbeddings :48-76, RobertaSelfAttention :79-126, RobertaSelfOutput :129-140, RobertaAttention :143-157,
RobertaIntermediate :160-169, RobertaOutput :172-183, RobertaLayer :186-198, RobertaEncoder :201-240,
RobertaModel :265-333, RobertaForCausalLMModified :336-406, RobertaLMHead :409-430, load_decoder :433-452).
Leaf modules are parameter containers; RobertaForCausalLMModified.forward runs the HIP layer program
(prismer_amd/programs/decoder.py).  transformers is not required: `config` is any object with the RobertaConfig
attribute names (a transformers.RobertaConfig works unchanged).
"""
import re
from types import SimpleNamespace

import torch
import torch.nn as nn

from ..config import PrismerDims
from ..programs.decoder import DecoderProgram
from ..store import ParamStore
from .utils import Adaptor, LayerNorm, _ContainerOnly

ROBERTA_PRETRAINED_MODEL_ARCHIVE_LIST = ['roberta-base', 'roberta-large']


class CausalLMOutput(SimpleNamespace):
    """stand-in for transformers' CausalLMOutputWithCrossAttentions: .loss [B], .logits [B,T,V]."""

    def __getitem__(self, i):
        return (self.loss, self.logits)[i] if self.loss is not None else (self.logits,)[i]


class RobertaEmbeddings(_ContainerOnly):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.register_buffer('position_ids', torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.padding_idx = config.pad_token_id
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size, padding_idx=self.padding_idx)


class RobertaSelfAttention(_ContainerOnly):
    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        kv_in = config.vision_hidden_size if is_cross_attention else config.hidden_size
        self.query = nn.Linear(config.hidden_size, config.hidden_size)
        self.key = nn.Linear(kv_in, config.hidden_size)
        self.value = nn.Linear(kv_in, config.hidden_size)


class RobertaSelfOutput(_ContainerOnly):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class RobertaAttention(_ContainerOnly):
    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.self = RobertaSelfAttention(config, is_cross_attention)
        self.output = RobertaSelfOutput(config)


class RobertaIntermediate(_ContainerOnly):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class RobertaOutput(_ContainerOnly):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class RobertaLayer(_ContainerOnly):
    def __init__(self, config):
        super().__init__()
        self.attention = RobertaAttention(config)
        self.intermediate = RobertaIntermediate(config)
        self.output = RobertaOutput(config)


class RobertaEncoder(_ContainerOnly):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([nn.ModuleList([RobertaLayer(config), RobertaAttention(config, is_cross_attention=True),
                                                   Adaptor(config.hidden_size, norm_late=True)]) for _ in range(config.num_hidden_layers)])
        self.output_layer = RobertaLayer(config)


class RobertaModel(_ContainerOnly):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = RobertaEmbeddings(config)
        self.encoder = RobertaEncoder(config)


class RobertaLMHead(_ContainerOnly):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.layer_norm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias                                   # roberta.py:417-419


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, input_ids, attention_mask, enc, labels, seed, *params):
        prog = mod._program()
        logits, loss, sv = prog.forward(input_ids, attention_mask, enc, labels, seed, save=True)
        ctx.mod, ctx.sv = mod, sv
        ctx.gradbuf = mod._store.begin_grads()
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        mod = ctx.mod
        st = mod._store
        st._grad_cur = ctx.gradbuf
        denc = mod._program().backward(ctx.sv, dloss)
        from .. import ops as _ops
        _ops.join_side()
        ctx.sv = None
        return (None, None, None, denc, None, None) + tuple(st.grads_for_autograd(mod._train_names))


class RobertaForCausalLMModified(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.roberta = RobertaModel(config)
        self.lm_head = RobertaLMHead(config)
        # tie_word_embeddings (transformers 4.26 post_init semantics, roberta.py:352-353)
        self.lm_head.decoder.weight = self.roberta.embeddings.word_embeddings.weight
        self._init_weights()
        self._store = None
        self._prog = None
        self._seed = None

    def _init_weights(self):
        """roberta.py:246-259."""
        std = getattr(self.config, 'initializer_range', 0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, std)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(0.0, std)
                if m.padding_idx is not None:
                    m.weight.data[m.padding_idx].zero_()
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_(); m.weight.data.fill_(1.0)

    def get_output_embeddings(self):
        return self.lm_head.decoder

    def dims(self) -> PrismerDims:
        c = self.config
        return PrismerDims(hidden_size=c.hidden_size, vision_hidden_size=c.vision_hidden_size, intermediate_size=c.intermediate_size,
                           num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads, vocab_size=c.vocab_size,
                           max_position_embeddings=c.max_position_embeddings, type_vocab_size=c.type_vocab_size,
                           pad_token_id=c.pad_token_id, layer_norm_eps=c.layer_norm_eps, hidden_dropout_prob=c.hidden_dropout_prob,
                           attention_probs_dropout_prob=c.attention_probs_dropout_prob)

    def _program(self):
        if self._store is None or not self._store.attached:
            self._store = ParamStore(self).attach()
            self._prog = None
        else:
            self._store.check_layout()
        if self._prog is None or self._layout_sig != self._store._freeze_sig:
            self._prog = DecoderProgram(self, self.dims(), self._store)
            self._layout_sig = self._store._freeze_sig
            self._train_names = [n for n in self._store.names if self._store.is_trainable(n)]
        return self._prog

    def dropout_seed(self):
        """device-resident dropout seed (advanced by ph_advance_seed after every training forward)."""
        if self._seed is None:
            dev = self.lm_head.bias.device
            self._seed = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=dev)
        return self._seed

    def forward(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, labels=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        from .. import ops
        if not input_ids.is_cuda:
            raise RuntimeError('prismer_amd.RobertaForCausalLMModified runs on the MI355X HIP path only.')
        prog = self._program()
        if not getattr(self._store, 'managed', False):
            self._store.refresh_if_stale()
        enc = encoder_hidden_states
        if enc.dtype != torch.bfloat16:
            enc = ops.cast_to_bf16(enc.contiguous().float())
        enc = enc.contiguous()
        B, T = input_ids.shape
        V = self.config.vocab_size
        seed = self.dropout_seed() if self.training else None
        if torch.is_grad_enabled() and labels is not None and (self._train_names or enc.requires_grad):
            params = [self._store.params[n] for n in self._train_names]
            # the backward regenerates the dropout masks from *seed when ITS kernels run, i.e. after the advance below:
            # this forward/backward pair gets its own snapshot of the seed (the persistent one moves on for the next forward)
            seed_fb = None if seed is None else seed.clone()
            loss, logits = _DecoderFn.apply(self, input_ids, attention_mask, enc, labels, seed_fb, *params)
        else:
            logits, loss, _ = prog.forward(input_ids, attention_mask, enc, labels, seed, save=False)
        if seed is not None:
            ops.advance_seed(seed)
        logits = logits.view(B, T, -1)[..., :V]
        if return_dict is False:
            return (loss, logits) if loss is not None else (logits,)
        return CausalLMOutput(loss=loss, logits=logits, hidden_states=None, attentions=None, cross_attentions=None)

    def decoding_program(self, encoder_hidden_states):
        """(layer program, encoder states as contiguous bf16) for KV-cached decoding (prismer_amd/model/generate.py)"""
        from .. import ops
        prog = self._program()
        if not getattr(self._store, 'managed', False):
            self._store.refresh_if_stale()
        enc = encoder_hidden_states
        if enc.dtype != torch.bfloat16:
            enc = ops.cast_to_bf16(enc.contiguous().float())
        return prog, enc.contiguous()

    @torch.no_grad()
    def generate(self, input_ids=None, encoder_hidden_states=None, attention_mask=None, num_beams=1, max_length=20, min_length=0,
                 length_penalty=1.0, **unused):
        """transformers' `generate` as the heads call it (model/prismer_caption.py:45-50, model/prismer_vqa.py:52-58): beam search,
        returns a [B, L] LongTensor padded with pad_token_id.  KV-cached on the HIP path (the reference re-runs the prefix)."""
        from ..model.generate import beam_search
        if attention_mask is None:
            attention_mask = input_ids.new_ones(input_ids.shape)
        c = self.config
        outs = beam_search(self, input_ids, attention_mask, encoder_hidden_states, num_beams=num_beams, max_length=max_length,
                           min_length=min_length, eos_token_id=getattr(c, 'eos_token_id', 2), pad_token_id=c.pad_token_id,
                           length_penalty=length_penalty)
        L = max(int(o.numel()) for o in outs)
        res = input_ids.new_full((len(outs), L), c.pad_token_id)
        for i, o in enumerate(outs):
            res[i, :o.numel()] = o
        return res

    def prepare_inputs_for_generation(self, input_ids, attention_mask=None, encoder_hidden_states=None, **kw):
        if attention_mask is None:
            attention_mask = input_ids.new_ones(input_ids.shape)
        return {'input_ids': input_ids, 'attention_mask': attention_mask, 'encoder_hidden_states': encoder_hidden_states}


def convert_roberta_state_dict(state_dict):
    """RobertaForMaskedLM checkpoint -> decoder keys: the surgery of roberta.py:440-447."""
    out = {}
    for key, v in state_dict.items():
        k = key
        if 'encoder.layer' in key:
            k = re.sub('.attention', '.0.attention', k)
            k = re.sub('.intermediate', '.0.intermediate', k)
            if 'attention' not in key:
                k = re.sub('.output', '.0.output', k)
        out[k] = v
    return out


def load_decoder(name: str, config, checkpoint_path: str = None):
    """Factory with the reference signature (roberta.py:433). No network here: weights come from `checkpoint_path`
    (a RobertaForMaskedLM state dict) when given, otherwise initializer_range random init."""
    if name not in ROBERTA_PRETRAINED_MODEL_ARCHIVE_LIST:
        raise RuntimeError(f'Model {name} not found')
    roberta = RobertaForCausalLMModified(config)
    if checkpoint_path is not None:
        roberta.load_state_dict(convert_roberta_state_dict(torch.load(checkpoint_path, map_location='cpu', weights_only=True)), strict=False)      # (a plain state dict: no unpickling of code)
    return roberta
