"""debug: Trainer capture with / without micro-batches on the tiny golden case."""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from golden import cases as C
import test_parity_gpu as T
from prismer_amd.trainer import Trainer
from prismer_amd import ops

mode = sys.argv[1]
case = C.Case('tiny_caption')
x, ids, mask, labels, _ = case.inputs()
enc, dec, esd, dsd = T.build(case, p_drop=0.0)
T.set_freeze(enc, dec)
class Holder(torch.nn.Module): pass
m = Holder(); m.expert_encoder, m.text_decoder = enc, dec
micro = 2 if 'micro' in mode else 1
tr = Trainer(m, lr=1e-3, total_steps=10, use_graph=False, micro_batches=micro)
tr.set_batch(T.to_dev(x), ids, mask, labels)
print('B =', ids.shape[0], 'micro =', tr.micro, flush=True)
for _ in range(2):
    tr.step()
torch.cuda.synchronize()
print('eager ok', float(tr.loss_buf), flush=True)
s = tr.static
which = mode.split(':')[1] if ':' in mode else 'all'
if 'nopool' in mode:
    ops.POOL = ops._NoPool()
g = torch.cuda.CUDAGraph()
if which == 'front':
    with torch.cuda.graph(g):
        for st in tr.stores: st.grad.zero_()
        h, xf, svf = tr.enc_prog.forward_front(s['experts'], tr.table, True, True)
elif which == 'trunk':
    h, xf, svf = tr.enc_prog.forward_front(s['experts'], tr.table, True, True)
    B = ids.shape[0]; d = tr.enc_prog.d; S, Mx = d.seq_len, d.num_expert_tokens
    with torch.cuda.graph(g):
        outs = []
        for mi, (b0, b1) in enumerate(tr._slices(B)):
            with ops.MICRO.branch(mi):
                outs.append(tr.enc_prog.forward_trunk(h[b0 * S:b1 * S], xf[b0 * Mx:b1 * Mx], b1 - b0, True))
        ops.MICRO.join()
elif which == 'decf':
    h, xf, svf = tr.enc_prog.forward_front(s['experts'], tr.table, True, True)
    B = ids.shape[0]; d = tr.enc_prog.d; S, Mx = d.seq_len, d.num_expert_tokens
    with torch.cuda.graph(g):
        outs = []
        for mi, (b0, b1) in enumerate(tr._slices(B)):
            with ops.MICRO.branch(mi):
                eo, svt = tr.enc_prog.forward_trunk(h[b0 * S:b1 * S], xf[b0 * Mx:b1 * Mx], b1 - b0, True)
                outs.append(tr.dec_prog.forward(s['input_ids'][b0:b1], s['attention_mask'][b0:b1], eo, s['labels'][b0:b1], tr.seed, True))
        ops.MICRO.join()
else:
    with torch.cuda.graph(g):
        tr._seg_forward_dec_backward(s)
print('capture ok', flush=True)
g.replay(); torch.cuda.synchronize()
print('replay ok', flush=True)
