import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.golden import cases as C
from tests.test_parity_gpu import build, set_freeze, to_dev
from prismer_amd.trainer import Trainer

case = C.Case('tiny_caption')
x, ids, mask, labels, _ = case.inputs()


class H(torch.nn.Module):
    pass


for use_graph, lr in ((False, 1e-3), (True, 1e-3), (False, 0.0), (True, 0.0), (False, 1e-3)):
    enc, dec, _, _ = build(case)
    set_freeze(enc, dec)
    dec._seed = torch.tensor([1234567], dtype=torch.int64, device='cuda')
    m = H(); m.expert_encoder, m.text_decoder = enc, dec
    tr = Trainer(m, lr=lr, total_steps=10, use_graph=use_graph, keep_grads=True)
    tr.set_batch(to_dev(x), ids, mask, labels)
    random.seed(99)
    orig = tr._host_prologue
    tabs = []
    def prologue():
        orig(); torch.cuda.synchronize(); tabs.append((tr.it, int(tr.table.sum().item()), round(float(tr.hyper[0].item()), 8), int(tr.seed.item()) % 100000))
    tr._host_prologue = prologue
    losses = [round(tr.step().item(), 5) for _ in range(2)]
    print('graph' if use_graph else 'eager', lr, losses, tabs)
