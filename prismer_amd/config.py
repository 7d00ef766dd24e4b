"""Model dimension tables for the Prismer hot path.

Mirrors the two sources the reference reads its dimensions from:
  * configs/prismer.json (reference: model/prismer.py:29-30) for the decoder, and
  * the CLIP checkpoint geometry inferred in model/modules/vit.py:211-214 for the ViT.
Nothing here touches the GPU.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List


# reference: model/prismer.py:18-27 -- expert name -> input channels
def expert_channels(experts) -> "OrderedDict[str, int]":
    out = OrderedDict()
    out['rgb'] = 3
    for exp in experts:
        if exp in ('depth', 'edge'):
            out[exp] = 1
        elif exp in ('normal',):
            out[exp] = 3
        elif 'seg' in exp:
            out['seg'] = 64
        elif exp in ('obj_detection', 'ocr_detection'):
            out[exp] = 64
    return out


LABEL_DOMAINS = ('seg', 'obj_detection', 'ocr_detection')   # vit.py:88
CAPTION_EXPERTS = ['depth', 'normal', 'seg_coco', 'edge', 'obj_detection', 'ocr_detection']  # configs/caption.yaml:5


@dataclass
class PrismerDims:
    # vision side (vit.py:78-131)
    image_resolution: int = 224
    patch_size: int = 16
    width: int = 768
    vit_layers: int = 12
    vit_heads: int = 12
    experts: "OrderedDict[str, int]" = field(default_factory=lambda: expert_channels(CAPTION_EXPERTS))
    expert_resolution: int = 224          # dataset/utils.py:43 -- expert maps are always 224x224
    resampler_layers: int = 4             # vit.py:129
    resampler_heads: int = 8
    num_latents: int = 64
    # language side (configs/prismer.json roberta_model)
    hidden_size: int = 768
    vision_hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    vocab_size: int = 50265
    max_position_embeddings: int = 514
    type_vocab_size: int = 1
    pad_token_id: int = 1
    bos_token_id: int = 0
    eos_token_id: int = 2
    layer_norm_eps: float = 1e-5
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    label_smoothing: float = 0.1          # roberta.py:384

    @property
    def rgb_grid(self) -> int:
        return self.image_resolution // self.patch_size

    @property
    def num_rgb_tokens(self) -> int:
        return self.rgb_grid ** 2

    @property
    def expert_grid(self) -> int:
        # dense stems: Upsample(16/p) then four stride-2 convs; label stems: Upsample(4/p) then two.
        up = int(self.expert_resolution * 16 / self.patch_size)
        g = up
        for _ in range(4):
            g = (g + 2 - 3) // 2 + 1
        return g

    @property
    def has_experts(self) -> bool:
        return len(self.experts) > 1

    @property
    def num_expert_tokens(self) -> int:
        return (len(self.experts) - 1) * self.expert_grid ** 2

    @property
    def seq_len(self) -> int:
        return self.num_rgb_tokens + (self.num_latents if self.has_experts else 0)

    def roberta_config_dict(self) -> Dict:
        return dict(attention_probs_dropout_prob=self.attention_probs_dropout_prob,
                    bos_token_id=self.bos_token_id, eos_token_id=self.eos_token_id,
                    hidden_act='gelu', hidden_dropout_prob=self.hidden_dropout_prob,
                    hidden_size=self.hidden_size, vision_hidden_size=self.vision_hidden_size,
                    initializer_range=0.02, intermediate_size=self.intermediate_size,
                    layer_norm_eps=self.layer_norm_eps, max_position_embeddings=self.max_position_embeddings,
                    num_attention_heads=self.num_attention_heads, num_hidden_layers=self.num_hidden_layers,
                    pad_token_id=self.pad_token_id, type_vocab_size=self.type_vocab_size,
                    vocab_size=self.vocab_size, is_decoder=True)


def prismer_base(experts: List[str] = None, image_resolution=224) -> PrismerDims:
    """configs/prismer.json 'prismer_base' + CLIP ViT-B/16 geometry."""
    ex = CAPTION_EXPERTS if experts is None else experts
    return PrismerDims(image_resolution=image_resolution, patch_size=16, width=768, vit_layers=12, vit_heads=12,
                       experts=expert_channels(ex))


def prismerz_base(image_resolution=224) -> PrismerDims:
    """PrismerZ: experts == 'none' -> rgb only (model/prismer.py:18-27)."""
    return prismer_base(experts=[], image_resolution=image_resolution)


def prismer_large(experts: List[str] = None, image_resolution=480) -> PrismerDims:
    """configs/prismer.json 'prismer_large' + CLIP ViT-L/14 geometry (vit.py:211-214: heads = width // 64)."""
    ex = CAPTION_EXPERTS if experts is None else experts
    return PrismerDims(image_resolution=image_resolution, patch_size=14, width=1024, vit_layers=24, vit_heads=16,
                       experts=expert_channels(ex), hidden_size=1024, vision_hidden_size=1024,
                       intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16)


def prismer_huge(experts: List[str] = None, image_resolution=224) -> PrismerDims:
    """configs/prismer.json 'prismer_huge' (:50-73: roberta-large decoder reading a 1280-wide vision stream) + CLIP ViT-H/14 geometry
    (vit.py:211-214: width 1280, 32 layers, heads = width // 64 = 20; the Experts Resampler keeps 8 heads -> head dim 160)."""
    ex = CAPTION_EXPERTS if experts is None else experts
    return PrismerDims(image_resolution=image_resolution, patch_size=14, width=1280, vit_layers=32, vit_heads=20,
                       experts=expert_channels(ex), hidden_size=1024, vision_hidden_size=1280,
                       intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16)


def prismer_tiny(experts: List[str] = None, image_resolution=64, expert_resolution=64, vocab_size=1003) -> PrismerDims:
    """Small geometry used by the golden fixtures (tests/golden/make_golden.py). Same code paths as BASE:
    7 stems, resampler (8 heads x 32), 2 ViT blocks (4 heads x 64), 2 decoder layers + output layer."""
    ex = CAPTION_EXPERTS if experts is None else experts
    return PrismerDims(image_resolution=image_resolution, patch_size=16, width=256, vit_layers=2, vit_heads=4,
                       experts=expert_channels(ex), expert_resolution=expert_resolution,
                       hidden_size=256, vision_hidden_size=256, intermediate_size=1024,
                       num_hidden_layers=2, num_attention_heads=4, vocab_size=vocab_size)


CONFIGS = {
    'prismer_base': prismer_base,
    'prismerz_base': prismerz_base,
    'prismer_large': prismer_large,
    'prismer_huge': prismer_huge,
    'prismer_tiny': prismer_tiny,
}
