"""Vision encoder shells (reference: model/modules/vit.py).

Same constructor signatures, attribute tree and state-dict keys as the reference classes
(ResidualAttentionBlock vit.py:37-59, Transformer :62-75, VisionTransformer :78-172, load_encoder :175-225), so
`model/prismer.py`, the freeze rule (prismer.py:39-59), FSDP wrap policies (train_caption.py:71-81) and strict
checkpoint loading keep working.  The forward/backward arithmetic is the HIP layer program
prismer_amd/programs/encoder.py; sub-modules are parameter containers only.
"""
import random
import re
from collections import OrderedDict

import torch
import torch.nn as nn

from ..config import PrismerDims, LABEL_DOMAINS
from ..programs.encoder import EncoderProgram
from ..store import ParamStore
from .resampler import PerceiverResampler
from .utils import Adaptor, LayerNorm, QuickGELU, _ContainerOnly, interpolate_pos_embed

# CLIP checkpoints the reference downloads (vit.py:28-34); here they must already be on disk (no network).
_MODELS = {'ViT-B/32': (768, 12, 12, 32), 'ViT-B/16': (768, 12, 12, 16), 'ViT-L/14': (1024, 24, 16, 14),
           'ViT-L/14@336px': (1024, 24, 16, 14), 'ViT-H/14': (1280, 32, 20, 14)}   # width, layers, heads(=width//64), patch


def _expert_map(v):
    """the tensor that carries an expert's spatial size: dense map, obj_detection {'label','instance'} (dataset/utils.py:149) or
    the compact {'label_map': uint8, 'table': fp32} form that is in-painted on the device, or {'raw': fp32} = a dense expert map before the
    min-max remap of post_label_process (remapped on the device, round 6)"""
    if isinstance(v, dict):
        return v['label_map'] if 'label_map' in v else (v['raw'] if 'raw' in v else v['label'])
    return v


class ResidualAttentionBlock(_ContainerOnly):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)       # container: in_proj_weight/bias, out_proj
        self.mlp = nn.Sequential(OrderedDict([('c_fc', nn.Linear(d_model, d_model * 4)), ('gelu', QuickGELU()),
                                              ('c_proj', nn.Linear(d_model * 4, d_model))]))
        self.ln_1 = LayerNorm(d_model)
        self.ln_2 = LayerNorm(d_model)


class Transformer(_ContainerOnly):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.resblocks = nn.Sequential(*[nn.ModuleList([ResidualAttentionBlock(width, heads), Adaptor(width)]) for _ in range(layers)])


def _stem(cin, width, label):
    s = (2, 2, 1, 1) if label else (2, 2, 2, 2)
    ch = [cin, width // 8, width // 4, width // 2, width]
    mods = [nn.Identity()]                                        # slot 0 = nn.UpsamplingBilinear2d in the reference (no params)
    for i in range(4):
        mods += [nn.Conv2d(ch[i], ch[i + 1], 3, stride=s[i], padding=1, bias=False), nn.BatchNorm2d(ch[i + 1]), nn.ReLU()]
    mods.append(nn.Conv2d(width, width, 1, bias=False))
    return nn.Sequential(*mods)


class _EncoderFn(torch.autograd.Function):
    """One autograd node for the whole encoder: forward/backward are the HIP layer programs."""

    @staticmethod
    def forward(ctx, mod, x, table, *params):
        prog = mod._program()
        out, sv = prog.forward(x, table, mod.training, save=True)
        ctx.mod, ctx.sv = mod, sv
        ctx.gradbuf = mod._store.begin_grads()
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        st = mod._store
        st._grad_cur = ctx.gradbuf
        mod._program().backward(ctx.sv, dout.contiguous())
        from .. import ops as _ops
        _ops.join_side()
        ctx.sv = None
        return (None, None, None) + tuple(st.grads_for_autograd(mod._train_names))


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, experts: dict):
        super().__init__()
        self.experts = experts
        self.input_resolution, self.patch_size, self.width, self.heads, self.layers = input_resolution, patch_size, width, heads, layers
        self.conv1 = nn.ModuleDict()
        for e in experts:
            if e == 'rgb':
                self.conv1[e] = nn.Conv2d(experts[e], width, kernel_size=patch_size, stride=patch_size, bias=False)
            else:
                label = e in LABEL_DOMAINS
                self.conv1[e] = _stem(64 if label else experts[e], width, label)
        scale = width ** -0.5
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2, width))
        if 'obj_detection' in self.experts:
            self.instance_embedding = nn.Parameter(scale * torch.randn(128, width))
        self.transformer = Transformer(width, layers, heads)
        if len(self.experts) > 1:
            self.resampler = PerceiverResampler(width=width, layers=4, heads=8, num_latents=64)
        self.ln_pre = LayerNorm(width)
        self.ln_post = LayerNorm(width)
        self.expert_resolution = 224                 # dataset/utils.py:43
        self.instance_mode = 'fast'                  # 'reference': draw per distinct label like vit.py:145-147 (host sync)
        self.instance_table = None                   # explicit int32[256] device table overrides both modes
        self._store = None
        self._prog = None

    # -------------------------------------------------------------------------------------------------
    def dims(self) -> PrismerDims:
        return PrismerDims(image_resolution=self.input_resolution, patch_size=self.patch_size, width=self.width, vit_layers=self.layers,
                           vit_heads=self.heads, experts=OrderedDict(self.experts), expert_resolution=self.expert_resolution)

    def _program(self):
        if self._store is None or not self._store.attached:
            self._store = ParamStore(self).attach()
            self._prog = None
        else:
            self._store.check_layout()
        if self._prog is None or self._prog.P is not self._store or self._layout_sig != self._store._freeze_sig:
            self._prog = EncoderProgram(self, self.dims(), self._store)
            self._layout_sig = self._store._freeze_sig
            self._train_names = [n for n in self._store.names if self._store.is_trainable(n)]
        return self._prog

    def _table(self, x):
        """label -> instance_embedding row.  Reference (vit.py:145-147): one random.randint(0,127) per DISTINCT
        instance id over the batch, drawn in sorted order (needs .unique(): device->host sync per step).
        'fast' draws 256 values up front from the same Python RNG: identical distribution, no sync."""
        if 'obj_detection' not in x:
            return None
        dev = x['rgb'].device
        if self.instance_table is not None:
            return self.instance_table
        if self.instance_mode == 'reference':
            t = [0] * 256
            for l in x['obj_detection']['instance'].unique().tolist():
                t[int(l) & 255] = random.randint(0, 127)
        else:
            t = [random.randint(0, 127) for _ in range(256)]
        return torch.tensor(t, dtype=torch.int32).to(dev, non_blocking=True)

    def forward(self, x: dict):
        if not x['rgb'].is_cuda:
            raise RuntimeError('prismer_amd.VisionTransformer runs on the MI355X HIP path only (inputs must be on the GPU).')
        for k, v in x.items():                       # expert maps are 224x224 in the reference pipeline (dataset/utils.py:43)
            if k != 'rgb':
                er = _expert_map(v).shape[-1]
                if er != self.expert_resolution:
                    self.expert_resolution, self._prog = er, None
                break
        prog = self._program()
        if not getattr(self._store, 'managed', False):
            self._store.refresh_if_stale()
        table = self._table(x)
        if torch.is_grad_enabled() and self._train_names:
            params = [self._store.params[n] for n in self._train_names]
            out = _EncoderFn.apply(self, x, table, *params)
        else:
            out, _ = prog.forward(x, table, self.training, save=False)
        return out.permute(1, 0, 2)                  # [S, B, D] like the reference (a view of the batch-major buffer)


def convert_clip_state_dict(state_dict):
    """CLIP checkpoint -> VisionTransformer keys: the surgery of vit.py:186-208 (visual.* only, drop `proj`, conv1 ->
    conv1.rgb, drop the class-token position, resblocks.{l}.x -> resblocks.{l}.0.x)."""
    out = {}
    for key, v in state_dict.items():
        if not key.startswith('visual'):
            continue
        k = key.replace('visual.', '')
        if 'proj' in k and 'transformer' not in k:
            continue
        if 'class_embedding' in k:
            out[k] = v                               # unexpected key, ignored by strict=False like the reference
        elif 'conv1' in k:
            out[k.replace('conv1', 'conv1.rgb')] = v
        elif 'positional_embedding' in k:
            out[k] = v[1:]
        elif 'transformer.resblocks' in k:
            k = re.sub('.mlp', '.0.mlp', k); k = re.sub('.attn', '.0.attn', k); k = re.sub('.ln', '.0.ln', k)
            out[k] = v
        else:
            out[k] = v
    return out


def load_encoder(name: str, experts: dict, image_resolution: int, checkpoint_path: str = None):
    """Factory with the reference signature (vit.py:175).  The reference downloads the CLIP weights; this
    environment has no network, so weights come from `checkpoint_path` (a CLIP state dict saved with torch.save) when
    given, otherwise the module keeps its random initialisation (synthetic-weight benchmarking)."""
    if name not in _MODELS:
        raise RuntimeError(f'Model {name} not found')
    width, layers, heads, patch = _MODELS[name]
    sd = None
    if checkpoint_path is not None:
        sd = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        sd = convert_clip_state_dict(sd.state_dict() if hasattr(sd, 'state_dict') else sd)
        # geometry comes from the checkpoint like in the reference (vit.py:211-214): width / patch from conv1, depth from the
        # number of attention blocks, heads = width // 64
        width, patch = sd['conv1.rgb.weight'].shape[0], sd['conv1.rgb.weight'].shape[-1]
        layers = len([k for k in sd if k.endswith('.attn.in_proj_weight')])
        heads = width // 64
    vit = VisionTransformer(input_resolution=image_resolution, patch_size=patch, width=width, layers=layers, heads=heads, experts=experts)
    if sd is not None:
        sd['positional_embedding'] = interpolate_pos_embed(sd['positional_embedding'], len(vit.positional_embedding))
        vit.load_state_dict(sd, strict=False)
    return vit
