"""Mint tests/golden/inpaint.npz by running the REFERENCE's post_label_process (dataset/utils.py:117-160) on synthetic label maps.

Build container only (`python -m tests.golden.make_inpaint_golden`): dataset/utils.py imports torchvision (absent here) at
module level -- only for the Transform class -- so torchvision is stubbed in sys.modules exactly like `clip` is for vit.py; the
function under test uses torch and the four feature files shipped in /root/reference/dataset only.  The fixture stores the label
maps, the labels_info side tables (as arrays), the feature tables the reference loaded (the GPU box has no reference files) and
the reference's painted outputs, sub-sampled every 3rd pixel per axis.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get('PRISMER_REFERENCE_ROOT', '/root/reference')
STRIDE = 3


def import_reference_utils():
    for name in ('torchvision', 'torchvision.transforms', 'torchvision.transforms.functional'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision.transforms'].functional = sys.modules['torchvision.transforms.functional']
    ra = types.ModuleType('dataset.randaugment'); ra.RandAugment = lambda *a, **k: None
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)                                   # dataset/utils.py loads 'dataset/*_features.pt' relative to the cwd
    try:
        import dataset                              # noqa: F401  (package __init__ imports the dataset classes)
    except Exception:
        pkg = types.ModuleType('dataset'); pkg.__path__ = [os.path.join(REF, 'dataset')]
        sys.modules['dataset'] = pkg
    sys.modules['dataset.randaugment'] = ra
    import importlib
    U = importlib.import_module('dataset.utils')
    os.chdir(cwd)
    return U


def synth_label_map(seed, E=224, n_labels=8, vocab=80):
    g = torch.Generator().manual_seed(seed)
    lab = torch.full((1, E, E), 255, dtype=torch.int64)
    ids = torch.randperm(vocab, generator=g)[:n_labels]
    for i in range(n_labels):
        y0, x0 = [int(v) for v in torch.randint(0, E - 8, (2,), generator=g)]
        h, w = [int(v) for v in torch.randint(4, E // 2, (2,), generator=g)]
        lab[0, y0:y0 + h, x0:x0 + w] = ids[i]
    return lab


def main():
    U = import_reference_utils()
    out = {'stride': np.int64(STRIDE)}
    out['feat.coco'] = U.COCO_FEATURES.float().numpy(); out['feat.ade'] = U.ADE_FEATURES.float().numpy()
    out['feat.det'] = U.DETECTION_FEATURES.float().numpy(); out['feat.bg'] = U.BACKGROUND_FEATURES.float().numpy()
    for img in range(2):
        g = torch.Generator().manual_seed(100 + img)
        seg = synth_label_map(10 + img, vocab=min(133, U.COCO_FEATURES.shape[0]))
        ade = synth_label_map(20 + img, vocab=min(150, U.ADE_FEATURES.shape[0]))
        obj = synth_label_map(30 + img, n_labels=6, vocab=6)          # instance ids 0..5 -> class ids through labels_info
        ocr = synth_label_map(40 + img, n_labels=4, vocab=4)
        obj_info = {str(i): int(torch.randint(0, U.DETECTION_FEATURES.shape[0], (1,), generator=g)) for i in range(6)}
        ocr_info = {i: {'features': torch.randn(64, generator=g) * 0.75} for i in range(4)}
        depth_in = torch.rand(1, 224, 224, generator=g) * 0.7 + 0.1
        # round 6: the other two dense experts -- a THREE-channel map (the min / max of dataset/utils.py:121 run over the whole [C, H, W] tensor, not
        # per channel) and an edge map; stored in full: the device-side remap (ph_dense_minmax_partial + the resize kernel) is tested on them
        normal_in = (torch.rand(3, 224, 224, generator=g) * torch.tensor([0.9, 0.5, 0.2]).view(3, 1, 1) + torch.tensor([0.05, 0.3, 0.6]).view(3, 1, 1)).half().float()
        edge_in = ((torch.rand(1, 224, 224, generator=g) > 0.93).float() * torch.rand(1, 224, 224, generator=g)).half().float()
        inputs = {'depth': depth_in.clone(), 'normal': normal_in.clone(), 'edge': edge_in.clone(), 'seg_coco': seg.clone(), 'seg_ade': ade.clone(),
                  'obj_detection': obj.clone(), 'ocr_detection': ocr.clone()}
        res = U.post_label_process(inputs, {'obj_detection': obj_info, 'ocr_detection': ocr_info})
        p = f'img{img}.'
        out[p + 'depth_in'] = depth_in.numpy()[:, ::STRIDE, ::STRIDE]
        out[p + 'depth_minmax'] = np.array([depth_in.min().item(), depth_in.max().item()])
        out[p + 'depth'] = res['depth'].numpy()[:, ::STRIDE, ::STRIDE]
        out[p + 'normal_in'], out[p + 'normal'] = normal_in.numpy().astype(np.float16), res['normal'].numpy()[:, ::STRIDE, ::STRIDE]
        out[p + 'edge_in'], out[p + 'edge'] = edge_in.numpy().astype(np.float16), res['edge'].numpy()[:, ::STRIDE, ::STRIDE]
        for k, lab in (('seg_coco', seg), ('seg_ade', ade), ('obj_detection', obj), ('ocr_detection', ocr)):
            out[p + k + '.map'] = lab.numpy().astype(np.uint8)
            r = res[k]['label'] if k == 'obj_detection' else res[k]
            out[p + k] = r.numpy()[:, ::STRIDE, ::STRIDE]
        assert torch.equal(res['obj_detection']['instance'], obj)
        out[p + 'obj_info'] = np.array([obj_info[str(i)] for i in range(6)], dtype=np.int64)
        out[p + 'ocr_feat'] = torch.stack([ocr_info[i]['features'] for i in range(4)]).numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'inpaint.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) / 1e6, 'MB')


if __name__ == '__main__':
    main()
