"""prismer_amd.optim.AdamW on parameters no ParamStore owns: the per-tensor path must be torch.optim.AdamW's arithmetic
(train_caption.py:111-112 builds the optimizer over `filter(requires_grad, model.parameters())`)."""
import copy

import torch

from prismer_amd.optim import AdamW


def _net():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.GELU(), torch.nn.Linear(13, 5))


def test_plain_path_equals_torch_adamw_and_state_dict_round_trips():
    a, b = _net(), None
    b = copy.deepcopy(a)
    oa = torch.optim.AdamW(a.parameters(), lr=3e-3, weight_decay=0.05)
    ob = AdamW(b.parameters(), lr=3e-3, weight_decay=0.05)
    x, y = torch.randn(11, 7), torch.randn(11, 5)
    for it in range(6):
        for g in (oa.param_groups + ob.param_groups):
            g['lr'] = 3e-3 * (1 - it / 10)                   # (the reference changes lr every step: utils.cosine_lr_schedule)
        for m, o in ((a, oa), (b, ob)):
            loss = ((m(x) - y) ** 2).mean()
            o.zero_grad(); loss.backward(); o.step()
        if it == 2:                                          # resume in a fresh optimizer mid-run
            ob2 = AdamW(b.parameters(), lr=1.0, weight_decay=0.05)
            ob2.load_state_dict(ob.state_dict())
            ob = ob2
    assert ob.fused_launches == 0 and ob.plain_updates == 4
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (p - q).abs().max()


def test_rejects_bad_hyperparameters_and_skips_missing_grads():
    m = _net()
    for kw in (dict(lr=-1.0), dict(eps=-1e-8), dict(betas=(1.0, 0.9)), dict(weight_decay=-0.1)):
        try:
            AdamW(m.parameters(), **kw)
        except ValueError:
            continue
        raise AssertionError(kw)
    o = AdamW(m.parameters())
    before = [p.detach().clone() for p in m.parameters()]
    o.step()                                                 # no gradients yet: nothing moves, no state is created
    assert all(torch.equal(p, q) for p, q in zip(m.parameters(), before)) and o.plain_updates == 0
