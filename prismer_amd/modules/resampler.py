"""Experts Resampler parameter tree (reference: model/modules/resampler.py:15-52).  The perceiver arithmetic runs in
prismer_amd/programs/encoder.py::EncoderProgram.resampler_fwd / resampler_bwd."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .utils import LayerNorm, SquaredReLU, _ContainerOnly


class PerceiverAttentionBlock(_ContainerOnly):
    def __init__(self, d_model: int, n_heads: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_heads)       # container for in_proj_* / out_proj.*
        self.mlp = nn.Sequential(OrderedDict([('c_fc', nn.Linear(d_model, d_model * 4)), ('sq_relu', SquaredReLU()),
                                              ('c_proj', nn.Linear(d_model * 4, d_model))]))
        self.ln_1 = LayerNorm(d_model)
        self.ln_2 = LayerNorm(d_model)
        self.ln_ff = LayerNorm(d_model)


class PerceiverResampler(_ContainerOnly):
    def __init__(self, width: int, layers: int, heads: int, num_latents: int):
        super().__init__()
        self.heads = heads
        self.latents = nn.Parameter(width ** -0.5 * torch.randn(num_latents, width))
        self.perceiver_blocks = nn.Sequential(*[PerceiverAttentionBlock(width, heads) for _ in range(layers)])
