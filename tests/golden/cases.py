"""Golden-case definitions shared by make_golden.py (runs the REFERENCE, build container only) and the tests
(which regenerate the identical weights / inputs from prismer_amd.synth and compare against the stored
reference outputs)."""
import random
from collections import OrderedDict

import torch

from prismer_amd import config, synth

INSTANCE_SEED = 7     # random.seed before every encoder call: vit.py:145-147 draws from Python's `random`


def _dims(name):
    if name in ('tiny_caption', 'tiny_vqa'):
        return config.prismer_tiny()
    if name == 'tiny_bicubic':            # rgb grid 6x6 vs expert grid 4x4 -> bicubic pos-embed path (utils.py:34-44)
        return config.prismer_tiny(image_resolution=96, expert_resolution=64)
    if name == 'tiny_z':                  # PrismerZ: rgb only, no resampler (vit.py:125,128)
        return config.prismer_tiny(experts=[])
    if name == 'base_caption':
        return config.prismer_base()
    raise KeyError(name)


CASES = OrderedDict([
    # name            : (batch, T, ragged)
    ('tiny_caption', (2, 12, True)),
    ('tiny_vqa', (3, 14, True)),
    ('tiny_bicubic', (2, 10, False)),
    ('tiny_z', (2, 12, True)),
    ('base_caption', (1, 30, False)),
])

LOGIT_STRIDE = {'base_caption': 97}       # store every 97th vocab column for the big case


class Case:
    def __init__(self, name):
        self.name = name
        self.dims = _dims(name)
        self.batch, self.T, self.ragged = CASES[name]
        self.seed = 0

    def weights(self):
        return synth.synth_encoder_state(self.dims, self.seed), synth.synth_decoder_state(self.dims, self.seed)

    def inputs(self):
        d = self.dims
        x = synth.synth_experts(d, self.batch, seed=1234)
        prompt = 4
        if self.name == 'tiny_vqa':       # prismer_vqa.py:32-33: everything but the answer span is -100
            ids, mask, labels = synth.synth_text(d, self.batch, self.T, seed=1234, ragged=True, prompt_length=1)
            ans = 4
            for b in range(self.batch):
                L = int(mask[b].sum())
                labels[b, :max(0, L - ans)] = -100
            weights = 0.6 + 0.4 * synth.uniform_pm1('in.vqa_w', (self.batch,), 1234)
            return x, ids, mask, labels, weights
        ids, mask, labels = synth.synth_text(d, self.batch, self.T, seed=1234, ragged=self.ragged, prompt_length=prompt)
        return x, ids, mask, labels, None

    def instance_table(self, x):
        """The table the reference's per-label random.randint draws produce after random.seed(INSTANCE_SEED)."""
        if 'obj_detection' not in x:
            return None
        rng = random.Random(INSTANCE_SEED)
        table = [0] * 256
        for l in x['obj_detection']['instance'].unique().tolist():
            table[int(l)] = rng.randint(0, 127)
        return table


# parameters whose FULL gradient is stored (the rest: L2 norm + 16 sampled entries)
FULL_GRAD_KEYS = (
    'expert_encoder.resampler.latents',
    'expert_encoder.ln_pre.weight',
    'expert_encoder.conv1.depth.2.weight',
    'expert_encoder.transformer.resblocks.0.1.adaptor.down_proj.bias',
    'text_decoder.lm_head.dense.bias',
    'text_decoder.roberta.encoder.layer.0.1.self.key.bias',
    'text_decoder.roberta.embeddings.LayerNorm.weight',
)


def sample_idx(key, numel, n=16):
    return synth.randint('gidx.' + key, (n,), 0, numel, 0)
