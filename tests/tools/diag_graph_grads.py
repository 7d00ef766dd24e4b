"""diagnostic: Trainer eager vs hipGraph gradients per parameter (dropout 0, lr 0)"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.golden import cases as C
from tests.test_parity_gpu import build, set_freeze, to_dev
from prismer_amd.trainer import Trainer
from prismer_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else 'tiny_caption'
case = C.Case(name)
x, ids, mask, labels, w = case.inputs()
tab = case.instance_table(x)


class H(torch.nn.Module):
    pass


def run(use_graph, steps=1, **kw):
    enc, dec, _, _ = build(case, p_drop=0.0)
    set_freeze(enc, dec)
    m = H(); m.expert_encoder, m.text_decoder = enc, dec
    tr = Trainer(m, lr=0.0, weight_decay=0.0, total_steps=10, use_graph=use_graph, keep_grads=True, **kw)
    tr.set_batch(to_dev(x), ids, mask, labels, None if w is None else w.cuda())
    if tab is not None:
        orig = tr._host_prologue
        def prologue():
            orig(); tr.table.copy_(torch.tensor(tab, dtype=torch.int32))
        tr._host_prologue = prologue
    for _ in range(steps):
        loss = tr.step()
    torch.cuda.synchronize()
    out = {}
    for pref, st in (('enc.', tr.stores[0]), ('dec.', tr.stores[1])):
        for nm in st.names:
            if st.is_trainable(nm):
                out[pref + nm] = st.g(nm).detach().float().cpu().clone()
    return loss.item(), out


def report(tag, ge, other):
    rows = []
    for k in ge:
        if ge[k].norm() > 1e-6:
            a, b = other[k].flatten().double(), ge[k].flatten().double()
            rows.append((float((a - b).norm() / b.norm()), float(a.norm() / b.norm()), float((a @ b) / (a.norm() * b.norm() + 1e-30)), k))
    rows.sort(reverse=True)
    print(' ', tag, 'n > 1e-3:', sum(r[0] > 1e-3 for r in rows), 'of', len(rows))
    for r in rows[:6]:
        print('     err %.4f  norm ratio %.4f  cos %.4f  %s' % r)


le, ge = run(False)
print('eager loss', le)
for tag, kw, env in (('graph default', {}, {}), ('graph no side streams', dict(side_stream=False), {}),
                     ('graph adamw overlap off', {}, {'PRISMER_ADAMW_OVERLAP': '0'}), ('eager no side streams', dict(side_stream=False), {'eager': 1})):
    for k, v in env.items():
        os.environ[k] = str(v)
    if kw.get('side_stream') is False:
        ops.SIDE = None; ops.POOL = ops._NoPool()
    l1, g1 = run('eager' not in env, **kw)
    for k in env:
        os.environ.pop(k, None)
    print(tag, 'loss', l1)
    report(tag, ge, g1)
