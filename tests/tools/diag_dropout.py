"""diagnostic: Trainer eager / graph / autograd losses with dropout ON, same seed -- who disagrees with whom?"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.golden import cases as C
from tests.test_parity_gpu import build, set_freeze, to_dev
from prismer_amd.trainer import Trainer

case = C.Case('tiny_caption')
x, ids, mask, labels, _ = case.inputs()
tab = case.instance_table(x)


class H(torch.nn.Module):
    pass


def trainer_loss(use_graph, steps=2, **kw):
    enc, dec, _, _ = build(case)
    set_freeze(enc, dec)
    dec._seed = torch.tensor([424242], dtype=torch.int64, device='cuda')
    m = H(); m.expert_encoder, m.text_decoder = enc, dec
    tr = Trainer(m, lr=0.0, weight_decay=0.0, total_steps=10, use_graph=use_graph, keep_grads=True, **kw)
    tr.set_batch(to_dev(x), ids, mask, labels)
    orig = tr._host_prologue
    def prologue():
        orig(); tr.table.copy_(torch.tensor(tab, dtype=torch.int32))
    tr._host_prologue = prologue
    out = []
    for _ in range(steps):
        out.append(round(tr.step().item(), 5)); torch.cuda.synchronize()
    return out, int(tr.seed.item())


def autograd_loss():
    enc, dec, _, _ = build(case)
    set_freeze(enc, dec)
    enc.train(); dec.train()
    dec._seed = torch.tensor([424242], dtype=torch.int64, device='cuda')
    enc.instance_table = torch.tensor(tab, dtype=torch.int32).cuda()
    out = []
    for _ in range(2):
        e = enc(to_dev(x))
        o = dec(ids.cuda(), attention_mask=mask.cuda(), encoder_hidden_states=e.permute(1, 0, 2), labels=labels.cuda(), return_dict=True)
        out.append(round(o.loss.mean().item(), 5))
    return out, int(dec._seed.item())


print('autograd      ', autograd_loss())
print('autograd again', autograd_loss())
print('eager         ', trainer_loss(False))
print('eager again   ', trainer_loss(False))
print('graph         ', trainer_loss(True))
print('graph again   ', trainer_loss(True))
print('eager 1 step  ', trainer_loss(False, steps=1))
os.environ['PRISMER_EXPERIMENTAL_GRAPH_BRANCHES'] = '1'
print('eager side    ', trainer_loss(False, side_stream=True))
