"""diagnostic: per-parameter gradient error of the HIP path vs the golden fixture (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_parity_gpu import build, set_freeze, run, C, GOLD
from tests.util import rel_fro
name = sys.argv[1] if len(sys.argv) > 1 else 'tiny_caption'
g = np.load(os.path.join(GOLD, name + '.npz'))
case = C.Case(name)
enc, dec, _, _ = build(case)
set_freeze(enc, dec)
enc.train(); dec.eval()
x, ids, mask, labels, weights = case.inputs()
e, out, total = run(enc, dec, case, x, ids, mask, labels, weights)
total.backward()
named = dict([('expert_encoder.' + n, p) for n, p in enc.named_parameters()] + [('text_decoder.' + n, p) for n, p in dec.named_parameters()])
rows = []
for n in str(g['requires_grad']).split('\n'):
    gn = float(g['gnorm.' + n]); gr = named[n].grad
    idx = C.sample_idx(n, gr.numel())
    samp = gr.flatten()[idx.cuda()].float().cpu().numpy()
    es = np.linalg.norm(samp - g['gsamp.' + n]) / (np.linalg.norm(g['gsamp.' + n]) + 1e-30)
    full = rel_fro(gr, torch.from_numpy(g['gfull.' + n])) if 'gfull.' + n in g else -1
    rows.append((abs(gr.double().norm().item() - gn) / (gn + 1e-30), es, full, gn, n))
rows.sort(reverse=True)
for r in rows[:40]:
    print('norm_err %.3e samp_err %.3e full %.3e gnorm %.3e %s' % r)
print('...')
for r in rows[-5:]:
    print('norm_err %.3e samp_err %.3e full %.3e gnorm %.3e %s' % r)
