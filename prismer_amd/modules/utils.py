"""Parameter containers mirroring model/modules/utils.py of the reference (names = state-dict contract).

These classes hold parameters only: the arithmetic of LayerNorm (utils.py:14-19), QuickGELU (:23-25),
SquaredReLU (:28-30) and Adaptor (:48-65) is executed by the HIP layer programs (prismer_amd/programs).
Calling one of these leaf modules directly is a usage error and raises -- there is no eager fallback.
"""
from collections import OrderedDict

import torch
import torch.nn as nn


class _ContainerOnly(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f'{type(self).__name__} is a parameter container of the HIP path; run the owning '
                           'VisionTransformer / RobertaForCausalLMModified instead (no eager fallback).')


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        raise RuntimeError('LayerNorm is executed by ph_layernorm_fwd inside the layer programs (no eager fallback).')


class QuickGELU(_ContainerOnly):
    pass


class SquaredReLU(_ContainerOnly):
    pass


class Adaptor(_ContainerOnly):
    def __init__(self, embed_dim: int, norm_late=False):
        super().__init__()
        self.norm_late = norm_late
        self.adaptor = nn.Sequential(OrderedDict([('down_proj', nn.Linear(embed_dim, embed_dim)), ('sq_relu', SquaredReLU()),
                                                  ('up_proj', nn.Linear(embed_dim, embed_dim))]))
        self.adaptor_ln = LayerNorm(embed_dim)


def interpolate_pos_embed(orig_pos_embed, target_len):
    """Checkpoint-time re-gridding of the positional table (reference utils.py:34-44, used by vit.py:223 and
    train_caption.py:98-99).  One-off host-side weight surgery, not on the step path: plain torch."""
    import torch.nn.functional as F
    o = int(orig_pos_embed.shape[0] ** 0.5)
    n = int(target_len ** 0.5)
    if o == n:
        return orig_pos_embed
    g = orig_pos_embed.reshape(1, o, o, -1).permute(0, 3, 1, 2)
    g = F.interpolate(g.float(), size=(n, n), mode='bicubic', align_corners=False)
    return g.permute(0, 2, 3, 1).flatten(0, 2).to(orig_pos_embed.dtype)
