"""prismer_amd.optim.AdamW under the reference's own loop (train_caption.py:111-136): with the drop-in modules every trainable parameter of a
top module is a view of one flat master and its gradient a view of one flat buffer, so the optimizer must take ONE fused launch per store and
still follow torch.optim.AdamW's trajectory."""
import os

import numpy as np
import pytest
import torch

from tests.golden import cases as C
from tests.test_parity_gpu import _head, to_dev

pytestmark = pytest.mark.gpu


def _model():
    from prismer_amd.model.prismer_caption import PrismerCaption
    case = C.Case('tiny_caption')
    m = _head(PrismerCaption, case)
    x, ids, mask, _, _ = case.inputs()
    return m, to_dev(x), (ids, mask)


def _loop(m, opt, x, cap, steps, lrs):
    losses = []
    for it in range(steps):
        for g in opt.param_groups:
            g['lr'] = lrs[it]
        loss = m(x, caption=cap, prefix=4, train=True)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses


def test_fused_store_update_equals_torch_adamw_on_the_same_gradients():
    """five steps of the reference loop with the fused optimizer; a twin set of plain tensors takes torch.optim.AdamW steps on COPIES of the
    very same gradients (the model's backward has fp32 atomics, so two separate runs only agree to Adam's sign-of-noise sensitivity -- the twin
    isolates the optimizer arithmetic)."""
    from prismer_amd.optim import AdamW
    lrs = [1e-4 * (1 - i / 8) for i in range(5)]
    m, x, cap = _model()
    params = [p for p in m.parameters() if p.requires_grad]
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt = AdamW(params, lr=1e-4, weight_decay=0.05)
    ref = torch.optim.AdamW(twin, lr=1e-4, weight_decay=0.05)
    losses = []
    for it in range(5):
        for g in opt.param_groups + ref.param_groups:
            g['lr'] = lrs[it]
        loss = m(x, caption=cap, prefix=4, train=True)
        opt.zero_grad()
        loss.backward()
        for p, q in zip(params, twin):
            q.grad = p.grad.detach().clone()
        opt.step()
        ref.step()
        assert opt.fused_launches == 2 and opt.plain_updates == 0, (opt.fused_launches, opt.plain_updates)   # one launch per store (encoder, decoder)
        for p, q in zip(params, twin):
            assert (p - q).abs().max().item() <= 2e-6 * lrs[it] / 1e-4 + 1e-6 * q.abs().max().item(), (it, (p - q).abs().max().item())
        losses.append(loss.item())
    assert losses[-1] < 0.7 * losses[0], losses
    # the bf16 shadows the next forward reads are the ones the fused launch wrote: equal to a fresh cast of the masters, nothing is stale
    for st in (m.expert_encoder._store, m.text_decoder._store):
        assert torch.equal(st.shadow[:st.n_train], st.master[:st.n_train].bfloat16())


def test_reference_loop_trajectory_with_fused_optimizer_tracks_torch_adamw():
    from prismer_amd.optim import AdamW
    lrs = [1e-4 * (1 - i / 8) for i in range(5)]
    ma, x, cap = _model()
    mb, _, _ = _model()
    oa = torch.optim.AdamW([p for p in ma.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.05)
    ob = AdamW([p for p in mb.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.05)
    la = _loop(ma, oa, x, cap, 5, lrs)
    lb = _loop(mb, ob, x, cap, 5, lrs)
    assert la[0] == lb[0]
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-3 * abs(a), (la, lb)          # (same bar as the hipGraph trajectory test: Adam's first steps amplify gradient round-off)


def _close(params, twin, tag):
    for p, q in zip(params, twin):
        assert (p - q).abs().max().item() <= 2e-6 + 1e-6 * q.abs().max().item(), (tag, (p - q).abs().max().item())


def test_state_dict_resume_and_partial_coverage_fall_back_cleanly():
    from prismer_amd.optim import AdamW
    m, x, cap = _model()
    params = [p for p in m.parameters() if p.requires_grad]
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt, ref = AdamW(params, lr=1e-4, weight_decay=0.05), torch.optim.AdamW(twin, lr=1e-4, weight_decay=0.05)
    for it in range(4):
        if it == 2:                                            # resume in a fresh optimizer: per-parameter exp_avg / exp_avg_sq / step, torch's usual layout
            sd = opt.state_dict()
            assert len(sd['state']) == len(params) and all(float(v['step']) == 2.0 for v in sd['state'].values())
            opt = AdamW(params, lr=1.0, weight_decay=0.0)
            opt.load_state_dict(sd)
        loss = m(x, caption=cap, prefix=4, train=True)
        opt.zero_grad()
        loss.backward()
        for p, q in zip(params, twin):
            q.grad = p.grad.detach().clone()
        opt.step(); ref.step()
        assert opt.fused_launches == 2 and opt.plain_updates == 0, it
        _close(params, twin, it)
    # an optimizer over only SOME parameters of a store cannot use the store-wide launch: per-tensor updates, the rest untouched
    m, x, cap = _model()
    named = dict(m.named_parameters())
    names = [n for n, p in named.items() if p.requires_grad][::3]
    params = [named[n] for n in names]
    rest = {n: p.detach().clone() for n, p in named.items() if n not in names}
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt, ref = AdamW(params, lr=1e-4, weight_decay=0.05), torch.optim.AdamW(twin, lr=1e-4, weight_decay=0.05)
    for it in range(2):
        loss = m(x, caption=cap, prefix=4, train=True)
        m.zero_grad()
        loss.backward()
        for p, q in zip(params, twin):
            q.grad = p.grad.detach().clone()
        opt.step(); ref.step()
        assert opt.fused_launches == 0 and opt.plain_updates == len(names)
        _close(params, twin, it)
    assert all(torch.equal(named[n], v) for n, v in rest.items())
    # ... and the next forward sees the per-tensor updates (the store notices the in-place writes and re-casts its bf16 shadows)
    m(x, caption=cap, prefix=4, train=True)
    st = m.text_decoder._store
    assert torch.equal(st.shadow[:st.n_train], st.master[:st.n_train].bfloat16())


def test_reassigned_gradient_leaves_the_fused_path_and_comes_back_without_losing_state():
    """a user who replaces one Parameter.grad (it no longer aliases the store's flat buffer) gets per-tensor updates for that step, with the
    moments and the step count carried over in both directions"""
    from prismer_amd.optim import AdamW
    m, x, cap = _model()
    params = [p for p in m.parameters() if p.requires_grad]
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt, ref = AdamW(params, lr=1e-4, weight_decay=0.05), torch.optim.AdamW(twin, lr=1e-4, weight_decay=0.05)
    kinds = []
    for it in range(4):
        loss = m(x, caption=cap, prefix=4, train=True)
        opt.zero_grad()
        loss.backward()
        if it == 1:
            params[3].grad = params[3].grad.clone()
        for p, q in zip(params, twin):
            q.grad = p.grad.detach().clone()
        opt.step(); ref.step()
        kinds.append((opt.fused_launches, opt.plain_updates > 0))
        for p, q in zip(params, twin):
            assert (p - q).abs().max().item() <= 2e-6 + 1e-6 * q.abs().max().item(), (it, (p - q).abs().max().item())
    assert kinds[0] == (2, False) and kinds[1][0] == 1 and kinds[1][1] and kinds[2] == (2, False) and kinds[3] == (2, False), kinds
    assert all(float(opt.state[p]['step']) == 4.0 for p in params)
