"""Label-expert in-painting (reference dataset/utils.py:117-160):
  CPU : the oracle restatement (label_table + post_label_process + remap_dense) against outputs of the REFERENCE function
        (tests/golden/inpaint.npz, minted by tests/golden/make_inpaint_golden.py)
  GPU : ph_inpaint_resize_nhwc against the oracle (bit-exact: index gather + the resize kernel's own tap arithmetic), and the
        encoder fed with compact {label_map, table} experts against the same encoder fed with the dense in-painted maps."""
import os

import numpy as np
import pytest
import torch

from oracle import prismer_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'inpaint.npz')
KINDS = ('seg_coco', 'seg_ade', 'obj_detection', 'ocr_detection')


def tables_for(g, img):
    p = f'img{img}.'
    bg = torch.from_numpy(g['feat.bg'])
    obj_info = {str(i): int(v) for i, v in enumerate(g[p + 'obj_info'])}
    ocr_info = {i: {'features': torch.from_numpy(f)} for i, f in enumerate(g[p + 'ocr_feat'])}
    return {'seg_coco': O.label_table('seg_coco', None, torch.from_numpy(g['feat.coco']), bg),
            'seg_ade': O.label_table('seg_ade', None, torch.from_numpy(g['feat.ade']), bg),
            'obj_detection': O.label_table('obj_detection', obj_info, torch.from_numpy(g['feat.det']), bg),
            'ocr_detection': O.label_table('ocr_detection', ocr_info, None, bg)}


def test_oracle_matches_reference_post_label_process():
    g = np.load(GOLD)
    s = int(g['stride'])
    for img in range(2):
        t = tables_for(g, img)
        for k in KINDS:
            lab = torch.from_numpy(g[f'img{img}.{k}.map'].astype(np.int64))
            got = O.post_label_process(lab, t[k])[:, ::s, ::s]
            assert torch.equal(got, torch.from_numpy(g[f'img{img}.{k}'])), (img, k)       # a gather: bit-exact
        lo, hi = g[f'img{img}.depth_minmax']
        d = torch.from_numpy(g[f'img{img}.depth_in'])
        got = 2 * (d - float(lo)) / (float(hi) - float(lo) + 1e-6) - 1                         # remap_dense on the sub-sampled map
        assert torch.allclose(got, torch.from_numpy(g[f'img{img}.depth']), atol=1e-6)
        # round 6: a three-channel map (min / max over the WHOLE [C, H, W] tensor) and an edge map, raw inputs stored in full
        for k in ('normal', 'edge'):
            raw = torch.from_numpy(g[f'img{img}.{k}_in'].astype(np.float32))
            assert torch.equal(O.remap_dense(raw)[:, ::s, ::s], torch.from_numpy(g[f'img{img}.{k}'])), (img, k)      # same torch expression: bit-exact
    x = torch.rand(1, 8, 8)
    assert float(O.remap_dense(x).min()) == -1.0 and abs(float(O.remap_dense(x).max()) - 1.0) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('Hout', [56, 64])            # patch 16 (BASE) and patch 14 (LARGE) stems: 224 -> 56 / 64 (vit.py:89)
def test_inpaint_resize_kernel_matches_oracle(Hout):
    from prismer_amd import ops
    g = np.load(GOLD)
    maps, tabs = [], []
    for img in range(2):
        t = tables_for(g, img)
        for k in KINDS:
            maps.append(torch.from_numpy(g[f'img{img}.{k}.map'])[0]); tabs.append(t[k])
    lab = torch.stack(maps)                                  # [8, 224, 224] uint8
    tab = torch.stack(tabs)                                  # [8, 256, 64] per-image tables
    dense = torch.stack([O.post_label_process(m[None], t) for m, t in zip(maps, tabs)])      # [8, 64, 224, 224] fp32 (the reference's tensor)
    want = ops.resize_to_nhwc(dense.cuda(), Hout, Hout)      # the dense path the compact one replaces
    got = ops.inpaint_resize(lab.cuda(), tab.cuda(), Hout, Hout)
    assert got.shape == want.shape == (8, Hout, Hout, 64)
    # same taps, weights and expression as the dense path; the two kernels may contract a*b+c differently (1 bf16 ulp)
    assert (got.float() - want.float()).abs().max() <= 2.0 ** -7 * want.float().abs().max()
    assert (got != want).float().mean() < 1e-2
    ref = torch.nn.functional.interpolate(dense, size=(Hout, Hout), mode='bilinear', align_corners=True).permute(0, 2, 3, 1)
    assert (got.float().cpu() - ref).abs().max() <= 2.0 ** -8 * ref.abs().max() + 1e-6      # bf16 rounding of the fp32 interpolation
    shared = ops.inpaint_resize(lab[:2].cuda(), tab[0].cuda(), Hout, Hout)                  # one table for the whole batch (fixed vocabulary)
    assert torch.equal(shared[0], got[0])                    # same kernel, shared vs per-image table


@pytest.mark.gpu
def test_encoder_with_compact_label_experts_equals_dense():
    from prismer_amd import synth
    from tests.golden import cases as C
    from tests.test_parity_gpu import build, to_dev
    case = C.Case('tiny_caption')
    d = case.dims
    enc, _, _, _ = build(case)
    enc.eval()
    x = case.inputs()[0]
    E = d.expert_resolution
    B = x['rgb'].shape[0]
    gen = torch.Generator().manual_seed(5)
    compact, dense = dict(x), dict(x)
    for name in ('seg_coco', 'obj_detection', 'ocr_detection'):
        lab = torch.randint(0, 12, (B, E, E), generator=gen).to(torch.uint8)
        lab[:, : E // 3] = 255
        tab = torch.randn(B, 256, 64, generator=gen) * 0.75
        img = torch.stack([O.post_label_process(lab[b][None], tab[b]) for b in range(B)])
        if name == 'obj_detection':
            dense[name] = {'label': img, 'instance': lab.long().unsqueeze(1)}
            compact[name] = {'label_map': lab, 'table': tab}
        else:
            dense[name] = img
            compact[name] = {'label_map': lab.unsqueeze(1), 'table': tab}
    tab_i = case.instance_table(x) or [3] * 256
    enc.instance_table = torch.tensor([(7 * i) % 128 for i in range(256)], dtype=torch.int32).cuda()
    with torch.no_grad():
        a = enc(to_dev(dense))
        b = enc(to_dev(compact))
    assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('Hout', [14, 30])            # the dense stems resize 224 -> 16 * E / p (vit.py:106): 224 at BASE; other sizes exercise the taps
def test_dense_remap_on_device_matches_reference_function(Hout):
    """ph_dense_minmax_partial + ph_resize_remap_nchw_to_nhwc (round 6) against the REFERENCE's post_label_process outputs (tests/golden/inpaint.npz:
    depth-like one-channel, three-channel normal, sparse edge maps): the per-sample min / max, the remapped taps (checked through an identity
    resize: Hout = Hin reproduces the remapped map itself to bf16 rounding) and the fused remap + resize against resize(remap) of the dense path."""
    from prismer_amd import ops
    g = np.load(GOLD)
    s = int(g['stride'])
    for k in ('normal', 'edge'):
        raw = torch.stack([torch.from_numpy(g[f'img{img}.{k}_in'].astype(np.float32)) for img in range(2)]).cuda()       # [2, C, 224, 224]
        raw[1] = raw[1] * 0.5 + 0.25                                       # the two samples of the batch get DIFFERENT ranges: the remap is per sample
        want_full = torch.stack([O.remap_dense(raw[b].cpu()) for b in range(2)])
        ident = ops.remap_resize_to_nhwc(raw, 224, 224).permute(0, 3, 1, 2).float().cpu()
        assert (ident - want_full).abs().max() <= 2.0 ** -8 * 1.0 + 1e-6, k                                # bf16 rounding of values in [-1, 1]
        ref0 = torch.from_numpy(g[f'img0.{k}'])                                                              # the reference function's own output (sample 0)
        assert (ident[0][:, ::s, ::s] - ref0).abs().max() <= 2.0 ** -8 + 1e-6, k
        got = ops.remap_resize_to_nhwc(raw, Hout, Hout)
        want = ops.resize_to_nhwc(want_full.cuda(), Hout, Hout)            # the dense path: the host remaps (reference expression), the device resizes
        assert got.shape == want.shape == (2, Hout, Hout, raw.shape[1])
        assert (got.float() - want.float()).abs().max() <= 2.0 ** -7, k     # same taps; the remap's division may differ in the last fp32 bit
        assert (got != want).float().mean() < 2e-2, k


@pytest.mark.gpu
def test_encoder_with_raw_dense_experts_equals_remapped():
    """the encoder fed {'raw': map} dense experts (remapped on the device) against the same encoder fed the host-remapped maps"""
    from tests.golden import cases as C
    from tests.test_parity_gpu import build, to_dev
    case = C.Case('tiny_caption')
    enc, _, _, _ = build(case)
    enc.eval()
    x = case.inputs()[0]
    gen = torch.Generator().manual_seed(9)
    raw_form, dense = dict(x), dict(x)
    for name in ('depth', 'normal', 'edge'):
        if name not in x:
            continue
        raw = torch.rand(x[name].shape, generator=gen) * 3.0 + 0.5
        raw[0] *= 0.3
        raw_form[name] = {'raw': raw}
        dense[name] = torch.stack([O.remap_dense(raw[b]) for b in range(raw.shape[0])])
    assert any(isinstance(v, dict) and 'raw' in v for v in raw_form.values())
    enc.instance_table = torch.tensor([(7 * i) % 128 for i in range(256)], dtype=torch.int32).cuda()       # (pins the per-forward random instance draw, vit.py:145-147)
    with torch.no_grad():
        a = enc(to_dev(dense))
        b = enc(to_dev(raw_form))
    assert torch.equal(a, b)                                 # (the remapped taps and the resize are bit-identical on both paths)
