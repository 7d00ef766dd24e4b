"""Mint the golden fixtures by running the REFERENCE module classes (from /root/reference) on CPU fp32.

Run in the build container only:   python -m tests.golden.make_golden
Writes tests/golden/<case>.npz.  Weights and inputs are NOT stored: they are regenerated bit-identically
from prismer_amd/synth.py (integer-hash generator), so a fixture holds only what the reference computed:

  enc_eval      VisionTransformer.forward in eval mode (BN running stats)           [S,B,D]
  logits_eval   RobertaForCausalLMModified logits (eval)                            [B,T,V] (strided for base)
  loss_eval     per-sample loss [B]  (roberta.py:381-387)
  enc_train / loss_train / total_train    same with the encoder in train() mode (BatchNorm batch statistics,
                decoder dropout off) -- total = caption .mean() or VQA (weights*loss).mean()
  bn.*          running_mean / running_var / num_batches_tracked of every BatchNorm after ONE train forward
  gnorm.* gsamp.* gfull.*   gradients of `total_train` under the reference freeze rule 'freeze_vision'
                (model/prismer.py:39-59 executed by the reference's own Prismer.prepare_to_train)
  gproj.*       N_PROJ fixed random projections of every gradient tensor (cases.grad_projections): a full-tensor error estimate
  ac_rel.*      exact full-tensor relative error of the bf16-autocast gradient (yardstick for the projection estimate)
  requires_grad names joined by '\n'
  ac_samp.* ac_norm.* ac_enc_train   error of PyTorch's own bf16 autocast (same reference modules, CPU) w.r.t. the fp32
                values above: the noise yardstick for the bf16 HIP path
"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_harness as RH                     # noqa: E402
from tests.golden import cases as C                      # noqa: E402


def run_case(name):
    case = C.Case(name)
    d = case.dims
    esd, dsd = case.weights()
    x, ids, mask, labels, weights = case.inputs()
    enc, dec = RH.build_reference(d, esd, dsd)
    out = {}

    def fwd():
        random.seed(C.INSTANCE_SEED)
        e = enc(x)
        o = dec(ids, attention_mask=mask, encoder_hidden_states=e.permute(1, 0, 2), labels=labels, return_dict=True)
        return e, o

    with torch.no_grad():
        e, o = fwd()
    stride = C.LOGIT_STRIDE.get(name, 1)
    es = C.ENC_STRIDE.get(name, 1)
    out['enc_eval'] = e[..., ::es].numpy()
    out['logits_eval'] = o.logits[..., ::stride].numpy()
    out['loss_eval'] = o.loss.numpy()

    holder = RH.reference_freeze(enc, dec, 'freeze_vision')
    enc.train()          # BatchNorm batch statistics; the encoder has no dropout. dec stays eval (dropout off).
    e, o = fwd()
    total = (o.loss if weights is None else weights * o.loss).mean()
    total.backward()
    out['enc_train'] = e.detach()[..., ::es].numpy()
    out['loss_train'] = o.loss.detach().numpy()
    out['total_train'] = total.detach().numpy()
    for k, v in enc.state_dict().items():
        if 'running_' in k or 'num_batches' in k:
            out['bn.' + k] = v.numpy().copy()
    names = []
    for n, p in holder.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        names.append(n)
        g = p.grad.detach()
        out['gnorm.' + n] = np.float64(g.double().norm().item())
        out['gsamp.' + n] = g.flatten()[C.sample_idx(n, g.numel())].numpy()
        out['gproj.' + n] = C.grad_projections(n, g)
        if n in C.FULL_GRAD_KEYS:
            out['gfull.' + n] = g.numpy()
    out['requires_grad'] = np.array('\n'.join(names))
    ref_full = {n: p.grad.detach().clone() for n, p in holder.named_parameters() if p.requires_grad}
    # yardstick: what eager PyTorch's OWN bf16 autocast does to the same gradients on the same modules (CPU).
    # The HIP path (bf16 storage, fp32 accumulate) is held to max(6e-2, 2x this) per parameter in tests/test_parity_gpu.py.
    for p in holder.parameters():
        p.grad = None
    for k, v in case.weights()[0].items():                       # restore the BatchNorm running statistics
        if 'running_' in k or 'num_batches' in k:
            enc.state_dict()[k].copy_(v)
    with torch.autocast('cpu', dtype=torch.bfloat16):
        e2, o2 = fwd()
        total2 = (o2.loss.float() if weights is None else weights * o2.loss.float()).mean()
    total2.backward()
    out['ac_enc_train'] = np.float64(((e2.float() - e.detach()).norm() / e.detach().norm()).item())
    for n, p in holder.named_parameters():
        if not p.requires_grad:
            continue
        ref_s = out['gsamp.' + n]
        s = p.grad.detach().flatten()[C.sample_idx(n, p.numel())].float().numpy()
        out['ac_samp.' + n] = np.float64(np.linalg.norm(s - ref_s) / (np.linalg.norm(ref_s) + 1e-30))
        out['ac_norm.' + n] = np.float64(abs(p.grad.double().norm().item() - float(out['gnorm.' + n])) / (float(out['gnorm.' + n]) + 1e-30))
        # FULL-tensor error of autocast (exact, the reference gradient is still at hand): yardstick of the projection test
        out['ac_rel.' + n] = np.float64((p.grad.double() - ref_full[n].double()).norm().item() / (float(out['gnorm.' + n]) + 1e-30))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)  loss_eval={out["loss_eval"]}  '
          f'total_train={out["total_train"]}  trainable={len(names)}')


def run_drop_case(name):
    """<case>_drop.npz (round 5): the reference classes in FULL training mode -- BatchNorm batch statistics AND the decoder's dropout
    (configs/prismer.json: 0.1 / 0.1, the arithmetic every benchmarked step runs) -- with every nn.Dropout replaced by the masks
    libprismer_hip draws for C.DROP_SEED (step 1) and splitmix64(C.DROP_SEED) (step 2: the library advances its device seed after every
    step); oracle.ref_harness.patch_dropout + tests/util.LibraryDropout.  Keys: s<k>.total_train, s<k>.loss_train, s<k>.gnorm.* / gsamp.* /
    gproj.* and the autocast yardsticks s<k>.ac_* measured under the same masks."""
    from tests.util import LibraryDropout, splitmix64
    case = C.Case(name)
    d = case.dims
    esd, dsd = case.weights()
    x, ids, mask, labels, weights = case.inputs()
    enc, dec = RH.build_reference(d, esd, dsd)
    holder = RH.reference_freeze(enc, dec, 'freeze_vision')
    enc.train()
    out = {}
    cur = {}
    RH.patch_dropout(dec, lambda site, t: cur['drop'](site, t))

    def fwd():
        random.seed(C.INSTANCE_SEED)
        e = enc(x)
        return e, dec(ids, attention_mask=mask, encoder_hidden_states=e.permute(1, 0, 2), labels=labels, return_dict=True)

    def reset():
        for p in holder.parameters():
            p.grad = None
        for k, v in case.weights()[0].items():
            if 'running_' in k or 'num_batches' in k:
                enc.state_dict()[k].copy_(v)
    seeds = [C.DROP_SEED, splitmix64(C.DROP_SEED)]
    names, ref_full = [], {}
    for si, seed in enumerate(seeds, 1):
        reset()
        cur['drop'] = LibraryDropout(seed, d.hidden_dropout_prob, d.attention_probs_dropout_prob)
        e, o = fwd()
        total = (o.loss if weights is None else weights * o.loss).mean()
        total.backward()
        pre = f's{si}.'
        out[pre + 'loss_train'] = o.loss.detach().numpy()
        out[pre + 'total_train'] = total.detach().numpy()
        names = []
        for n, p in holder.named_parameters():
            if not p.requires_grad:
                continue
            names.append(n)
            g = p.grad.detach()
            out[pre + 'gnorm.' + n] = np.float64(g.double().norm().item())
            out[pre + 'gsamp.' + n] = g.flatten()[C.sample_idx(n, g.numel())].numpy()
            out[pre + 'gproj.' + n] = C.grad_projections(n, g)
        ref_full[si] = {n: p.grad.detach().clone() for n, p in holder.named_parameters() if p.requires_grad}
    out['requires_grad'] = np.array('\n'.join(names))
    out['seeds'] = np.array(seeds, dtype=np.uint64)
    for si, seed in enumerate(seeds, 1):                     # autocast yardsticks under the SAME masks, per step
        pre = f's{si}.'
        reset()
        cur['drop'] = LibraryDropout(seed, d.hidden_dropout_prob, d.attention_probs_dropout_prob)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            e2, o2 = fwd()
            total2 = (o2.loss.float() if weights is None else weights * o2.loss.float()).mean()
        total2.backward()
        for n, p in holder.named_parameters():
            if not p.requires_grad:
                continue
            ref_s = out[pre + 'gsamp.' + n]
            gn = float(out[pre + 'gnorm.' + n])
            sm = p.grad.detach().flatten()[C.sample_idx(n, p.numel())].float().numpy()
            out[pre + 'ac_samp.' + n] = np.float64(np.linalg.norm(sm - ref_s) / (np.linalg.norm(ref_s) + 1e-30))
            out[pre + 'ac_norm.' + n] = np.float64(abs(p.grad.double().norm().item() - gn) / (gn + 1e-30))
            out[pre + 'ac_rel.' + n] = np.float64((p.grad.double() - ref_full[si][n].double()).norm().item() / (gn + 1e-30))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name + '_drop.npz')
    np.savez_compressed(path, **out)
    print(f'{name}_drop: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)  total_train s1={out["s1.total_train"]} s2={out["s2.total_train"]}')


def run_traj_case(name):
    """<case>_traj.npz (round 6): C.TRAJ_STEPS iterations of the REFERENCE training loop (train_caption.py:111-112 torch.optim.AdamW over the
    requires_grad parameters, :126-135 cosine_lr_schedule -> forward -> zero_grad -> backward -> step) on the reference classes in full training
    mode (BatchNorm batch statistics, dropout 0.1 / 0.1 under the library's masks: step k draws with splitmix64^(k-1)(C.DROP_SEED), exactly what the
    Trainer's device seed does), same batch every step.  Keys: `losses` [steps], bn.* after the last step, per trainable tensor dnorm.* = |p_K - p_0|,
    dproj.* = the N_PROJ projections of p_K - p_0; ac_losses / ac_drel.*: the same loop under PyTorch's own bf16 autocast (the yardstick:
    AdamW's first steps are ~ lr * sign(g), so round-off level gradient entries flip and the update error is far above the gradient error)."""
    import math
    from tests.util import LibraryDropout, splitmix64
    case = C.Case(name)
    d = case.dims
    x, ids, mask, labels, weights = case.inputs()
    out = {}

    def loop(autocast):
        esd, dsd = case.weights()
        enc, dec = RH.build_reference(d, esd, dsd)
        holder = RH.reference_freeze(enc, dec, 'freeze_vision')
        enc.train()
        cur = {}
        RH.patch_dropout(dec, lambda site, t: cur['drop'](site, t))
        params = [(n, p) for n, p in holder.named_parameters() if p.requires_grad]
        p0 = {n: p.detach().clone() for n, p in params}
        opt = torch.optim.AdamW(params=[p for _, p in params], lr=C.TRAJ_LR, weight_decay=C.TRAJ_WD)      # train_caption.py:111-112
        seed, losses = C.DROP_SEED, []
        for it in range(C.TRAJ_STEPS):
            lr = (C.TRAJ_LR - C.TRAJ_MIN_LR) * 0.5 * (1. + math.cos(math.pi * it / C.TRAJ_TOTAL)) + C.TRAJ_MIN_LR     # utils.py:13-17
            for gp in opt.param_groups:
                gp['lr'] = lr
            cur['drop'] = LibraryDropout(seed, d.hidden_dropout_prob, d.attention_probs_dropout_prob)
            random.seed(C.INSTANCE_SEED)
            if autocast:
                with torch.autocast('cpu', dtype=torch.bfloat16):
                    e = enc(x)
                    o = dec(ids, attention_mask=mask, encoder_hidden_states=e.permute(1, 0, 2), labels=labels, return_dict=True)
                    loss = (o.loss.float() if weights is None else weights * o.loss.float()).mean()
            else:
                e = enc(x)
                o = dec(ids, attention_mask=mask, encoder_hidden_states=e.permute(1, 0, 2), labels=labels, return_dict=True)
                loss = (o.loss if weights is None else weights * o.loss).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            seed = splitmix64(seed)
            print(f'{name}_traj autocast={autocast} step {it + 1}: lr {lr:.3e} loss {losses[-1]:.6f}', flush=True)
        delta = {n: (p.detach() - p0[n]) for n, p in params}
        return losses, delta, enc

    losses, delta, enc = loop(False)
    out['losses'] = np.array(losses, dtype=np.float64)
    for k, v in enc.state_dict().items():
        if 'running_' in k or 'num_batches' in k:
            out['bn.' + k] = v.numpy().copy()
    for n, dl in delta.items():
        out['dnorm.' + n] = np.float64(dl.double().norm().item())
        out['dproj.' + n] = C.grad_projections(n, dl)
    out['requires_grad'] = np.array('\n'.join(delta))
    ac_losses, ac_delta, _ = loop(True)
    out['ac_losses'] = np.array(ac_losses, dtype=np.float64)
    for n, dl in ac_delta.items():
        out['ac_drel.' + n] = np.float64((dl.double() - delta[n].double()).norm().item() / (float(out['dnorm.' + n]) + 1e-30))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name + '_traj.npz')
    np.savez_compressed(path, **out)
    print(f'{name}_traj: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)  losses={losses}  autocast losses={ac_losses}')


if __name__ == '__main__':
    assert RH.available(), 'reference not mounted'
    torch.set_num_threads(int(os.environ.get('GOLDEN_THREADS', os.cpu_count())))
    for name in (sys.argv[1:] or list(C.CASES) + [n + '+drop' for n in C.DROP_CASES] + [n + '+traj' for n in C.TRAJ_CASES]):
        if name.endswith('+drop'):
            run_drop_case(name[:-5])
        elif name.endswith('+traj'):
            run_traj_case(name[:-5])
        else:
            run_case(name)
