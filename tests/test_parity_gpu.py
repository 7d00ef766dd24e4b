"""GPU: the HIP path (through the C ABI and the nn.Module shells) against
  (1) the committed REFERENCE outputs (tests/golden/*.npz, produced by the reference module classes), and
  (2) the CPU oracle (oracle/prismer_oracle.py) on the same synthetic weights/inputs.
Tolerances follow SURVEY 8c (bf16 storage / fp32 accumulate vs an fp32 reference):
  encoder output, logits  rel-Frobenius <= 2e-2 ; per-sample loss rel <= 2e-3 ; gradients rel-Frobenius <= 6e-2."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import prismer_oracle as O
from prismer_amd import config, synth
from prismer_amd.model.prismer import _Cfg
from prismer_amd.modules.roberta import RobertaForCausalLMModified
from prismer_amd.modules.vit import VisionTransformer
from tests.golden import cases as C
from tests.util import rel_fro

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
TOL_ACT, TOL_LOSS, TOL_GRAD = 2e-2, 2e-3, 5e-2      # (gradient bar 5e-2 since round 6: SURVEY 8c; 6e-2 before)
TOL_GRAD_SAMPLED = 6e-2      # the 16-ENTRY sample is a noisy estimator of a tensor's error (B = 1 fixtures sit at 0.84 of 2 x this): it keeps the round-2 bar;
                             # the full-tensor checks (norm, 16 projections, gfull) carry the tightened one


def to_dev(x):
    if isinstance(x, dict):
        return {k: to_dev(v) for k, v in x.items()}
    return x.cuda()


def build(case, p_drop=None):
    d = case.dims
    if p_drop is not None:
        d.hidden_dropout_prob = d.attention_probs_dropout_prob = p_drop
    enc = VisionTransformer(d.image_resolution, d.patch_size, d.width, d.vit_layers, d.vit_heads, dict(d.experts))
    dec = RobertaForCausalLMModified(_Cfg(d.roberta_config_dict()))
    esd, dsd = case.weights()
    enc.load_state_dict(esd, strict=True)
    dec.load_state_dict(dsd, strict=True)
    return enc.cuda(), dec.cuda(), esd, dsd


def set_freeze(enc, dec, mode='freeze_vision'):
    for n, p in enc.named_parameters():
        p.requires_grad = not ('transformer.resblocks' in ('expert_encoder.' + n) and 'adaptor' not in n) if mode == 'freeze_vision' else True
    for n, p in dec.named_parameters():
        p.requires_grad = True


def run(enc, dec, case, x, ids, mask, labels, weights):
    tab = case.instance_table(x)
    enc.instance_table = None if tab is None else torch.tensor(tab, dtype=torch.int32).cuda()
    e = enc(to_dev(x))
    out = dec(ids.cuda(), attention_mask=mask.cuda(), encoder_hidden_states=e.permute(1, 0, 2), labels=labels.cuda(), return_dict=True)
    total = (out.loss if weights is None else weights.cuda() * out.loss).mean()
    return e, out, total


@pytest.mark.parametrize('name', list(C.CASES))
def test_eval_matches_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    case = C.Case(name)
    enc, dec, _, _ = build(case)
    enc.eval(); dec.eval()
    x, ids, mask, labels, weights = case.inputs()
    with torch.no_grad():
        e, out, _ = run(enc, dec, case, x, ids, mask, labels, weights)
    s = C.LOGIT_STRIDE.get(name, 1)
    r_enc = rel_fro(e.float()[..., ::C.ENC_STRIDE.get(name, 1)], torch.from_numpy(g['enc_eval']))
    r_log = rel_fro(out.logits.float()[..., ::s], torch.from_numpy(g['logits_eval']))
    r_loss = rel_fro(out.loss, torch.from_numpy(g['loss_eval']))
    print(f'{name}: enc {r_enc:.2e} logits {r_log:.2e} loss {r_loss:.2e}')
    assert r_enc < TOL_ACT and r_log < TOL_ACT and r_loss < TOL_LOSS, (r_enc, r_log, r_loss)


@pytest.mark.parametrize('name', ['tiny_caption', 'tiny_vqa', 'tiny_bicubic', 'tiny_z', 'tiny_vqa_head', 'base_caption', 'base_b8'])
def test_train_mode_and_gradients_match_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    case = C.Case(name)
    enc, dec, _, _ = build(case)
    set_freeze(enc, dec)
    enc.train(); dec.eval()            # BatchNorm batch statistics, decoder dropout off (= how the fixture was minted)
    x, ids, mask, labels, weights = case.inputs()
    e, out, total = run(enc, dec, case, x, ids, mask, labels, weights)
    total.backward()
    assert rel_fro(e.float()[..., ::C.ENC_STRIDE.get(name, 1)], torch.from_numpy(g['enc_train'])) < TOL_ACT
    assert rel_fro(out.loss, torch.from_numpy(g['loss_train'])) < TOL_LOSS
    for k, v in enc.state_dict().items():                        # running-stat update of every BatchNorm (App. C #7)
        if 'running_' in k:
            assert rel_fro(v, torch.from_numpy(g['bn.' + k])) < 5e-3, k
        if 'num_batches' in k:
            assert int(v) == int(g['bn.' + k])
    trainable = str(g['requires_grad']).split('\n')
    named = dict([('expert_encoder.' + n, p) for n, p in enc.named_parameters()] + [('text_decoder.' + n, p) for n, p in dec.named_parameters()])
    got = sorted(n for n, p in named.items() if p.requires_grad)
    assert got == sorted(trainable)
    # Gradient bar: per parameter, error <= max(TOL_GRAD, 2 x the error PyTorch's OWN bf16 autocast makes on the reference
    # modules for that parameter) -- the fixture stores that yardstick (ac_samp / ac_norm).  The expert stems see
    # 15-70 % element-wise bf16 noise under autocast (train-mode BatchNorm over few, piecewise-constant samples);
    # everything else sits at 1-3 %.
    worst = []
    for n in trainable:
        gn = float(g['gnorm.' + n])
        gr = named[n].grad
        assert gr is not None, n
        if gn < 1e-4:
            continue                                            # analytically-zero gradients (attention key biases)
        e_norm = abs(gr.double().norm().item() - gn) / gn
        idx = C.sample_idx(n, gr.numel())
        samp = gr.flatten()[idx.cuda()].float().cpu().numpy()
        e_samp = float(np.linalg.norm(samp - g['gsamp.' + n]) / (np.linalg.norm(g['gsamp.' + n]) + 1e-30))
        bar_s = max(2 * TOL_GRAD_SAMPLED, 2.0 * float(g['ac_samp.' + n]))    # 16 sampled entries: 2x head-room on the element-wise bar
        bar_n = max(TOL_GRAD, 2.0 * float(g['ac_norm.' + n]))
        worst.append((e_samp / bar_s, e_norm / bar_n, e_samp, e_norm, n))
        if 'gfull.' + n in g:
            r = rel_fro(gr, torch.from_numpy(g['gfull.' + n]))
            assert r < max(bar_s, TOL_GRAD), (n, r, bar_s)
    worst.sort(reverse=True)
    print(name, 'worst (samp/bar, norm/bar, samp, norm, name):', worst[:4])
    med = float(np.median([w[2] for w in worst]))
    print(name, 'median sampled-gradient error', med)
    assert worst[0][0] < 1.0 and max(w[1] for w in worst) < 1.0, worst[:4]
    assert med < 3e-2


def test_trainer_step_matches_oracle_adamw():
    """Native Trainer (no autograd, fused AdamW, hipGraph off and on) vs oracle forward/backward + AdamW formula."""
    from prismer_amd.trainer import Trainer, cosine_lr
    case = C.Case('tiny_caption')
    d = case.dims
    x, ids, mask, labels, _ = case.inputs()
    tab = case.instance_table(x)

    class Holder(torch.nn.Module):
        pass

    def make():
        enc, dec, esd, dsd = build(case, p_drop=0.0)
        set_freeze(enc, dec)
        m = Holder(); m.expert_encoder, m.text_decoder = enc, dec
        return m, esd, dsd
    results = []
    for use_graph in (False, True):
        m, esd, dsd = make()
        tr = Trainer(m, lr=1e-3, total_steps=10, use_graph=use_graph, keep_grads=True)
        tr.set_batch(to_dev(x), ids, mask, labels)
        random.seed(0)
        m.expert_encoder.instance_table = None
        orig_prologue = tr._host_prologue

        def prologue():
            orig_prologue()
            tr.table.copy_(torch.tensor(tab, dtype=torch.int32))
        tr._host_prologue = prologue
        loss = tr.step()
        torch.cuda.synchronize()
        if not use_graph:
            grads_eager = {}
            for pref, st in (('expert_encoder.', tr.stores[0]), ('text_decoder.', tr.stores[1])):
                for nm in st.names:
                    if st.is_trainable(nm):
                        grads_eager[pref + nm] = st.g(nm).detach().float().cpu().clone()
        results.append((loss.item(), {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}))
    # oracle: one step
    esd, dsd = case.weights()
    names = ['expert_encoder.' + k for k in esd] + ['text_decoder.' + k for k in dsd]
    fm = O.freeze_mask(names, 'freeze_vision')
    leaves = {}
    for k, v in esd.items():
        if v.is_floating_point() and 'running' not in k and fm['expert_encoder.' + k]:
            v.requires_grad_(True); leaves['expert_encoder.' + k] = v
    for k, v in dsd.items():
        if v.is_floating_point() and not k.startswith('lm_head.decoder.') and fm['text_decoder.' + k]:
            v.requires_grad_(True); leaves['text_decoder.' + k] = v
    eo = O.vision_encoder(esd, x, d.patch_size, d.vit_heads, True, tab, {})
    _, ls = O.text_decoder(dsd, ids, mask, eo.transpose(0, 1), d.num_attention_heads, labels)
    ls.mean().backward()
    lr = cosine_lr(0, 10, 1e-3, 0.0)
    for use_graph, (loss, sd) in zip((False, True), results):
        # capture warm-up is side-effect free (round 2): the first replayed step is the first step, like the eager one
        assert math_close(loss, ls.mean().item(), 2e-3), (use_graph, loss, ls.mean().item())
    loss, sd = results[0]
    g = np.load(os.path.join(GOLD, 'tiny_caption.npz'))
    # (1) gradients left in the flat fp32 buffers by the hand-scheduled backward == oracle autograd gradients
    errs = []
    for n, v in leaves.items():
        gn = float(g['gnorm.' + n])
        if gn < 1e-4:
            continue
        got = grads_eager[n]
        e_norm = abs(got.double().norm().item() - v.grad.double().norm().item()) / v.grad.double().norm().item()
        errs.append((e_norm / max(TOL_GRAD, 2 * float(g['ac_norm.' + n])), n))
    errs.sort(reverse=True)
    print('trainer worst grad-norm error / bar', errs[:3])
    assert errs[0][0] < 1.0
    # (2) the fused AdamW moved every trainable parameter by ~lr (first Adam step: |delta| = lr*|g|/(|g|+eps) + lr*wd*|p|)
    for n, v in leaves.items():
        if float(g['gnorm.' + n]) < 1e-4:
            continue
        delta = (sd[n] - v.detach()).abs()
        assert delta.max() <= lr * (1.0 + 0.05 * v.detach().abs().max().item()) * 1.01 + 1e-7, n
        if (v.grad != 0).float().mean() > 0.9:               # embedding tables: most rows get no gradient
            assert delta.mean() > 0.5 * lr, n
    # (3) frozen parameters and the hipGraph path
    assert torch.equal(sd['expert_encoder.transformer.resblocks.0.0.attn.in_proj_weight'],
                       case.weights()[0]['transformer.resblocks.0.0.attn.in_proj_weight'])
    assert np.isfinite(results[1][0])


def test_trainer_micro_batch_branches_match_whole_batch():
    """Trainer(micro_batches=2, side_stream=True): stems on the whole batch (BatchNorm statistics), trunk + decoder as two parallel
    branches (eager launches on extra streams).  Same loss and the same gradients as the single-slice schedule (dropout off)."""
    from prismer_amd.trainer import Trainer
    case = C.Case('tiny_caption')
    x, ids, mask, labels, _ = case.inputs()

    def rep(t):                                   # batch 2 -> 4: two slices of two images
        return {k: rep(v) for k, v in t.items()} if isinstance(t, dict) else torch.cat([t, t.flip(0)], 0)
    x4, ids4, mask4, labels4 = rep(x), rep(ids), rep(mask), rep(labels)
    tab = case.instance_table(x)

    class Holder(torch.nn.Module):
        pass
    grads, losses = {}, {}
    for micro, use_graph in ((1, False), (2, False)):
        enc, dec, _, _ = build(case, p_drop=0.0)
        set_freeze(enc, dec)
        m = Holder(); m.expert_encoder, m.text_decoder = enc, dec
        tr = Trainer(m, lr=0.0, weight_decay=0.0, total_steps=10, use_graph=use_graph, micro_batches=micro, keep_grads=True, side_stream=micro > 1)
        tr.set_batch(to_dev(x4), ids4, mask4, labels4)
        assert len(tr._slices(4)) == micro
        m.expert_encoder.instance_table = None
        orig_prologue = tr._host_prologue

        def prologue(tr=tr, orig=orig_prologue):  # pin the instance-embedding draw (Python RNG) for every step
            orig()
            tr.table.copy_(torch.tensor(tab, dtype=torch.int32))
        tr._host_prologue = prologue
        loss = tr.step()                          # lr = 0: warm-up steps of the capture do not move the parameters
        torch.cuda.synchronize()
        losses[(micro, use_graph)] = loss.item()
        grads[(micro, use_graph)] = torch.cat([st.grad[:st.n_train].float().cpu() for st in tr.stores])
    ref = grads[(1, False)]
    for k in ((2, False),):
        assert math_close(losses[k], losses[(1, False)], 1e-4), (k, losses)
        err = (grads[k] - ref).norm() / ref.norm()
        assert err < 2e-3, (k, float(err))       # same kernels on half-size problems: accumulation-order noise only


def math_close(a, b, rel):
    return abs(a - b) <= rel * abs(b)


def test_dropout_training_forward_is_reproducible_and_unbiased():
    """decoder dropout (p=0.1) active: same seed -> identical loss; different seeds -> losses scatter around eval loss."""
    case = C.Case('tiny_caption')
    enc, dec, _, _ = build(case)
    enc.eval(); dec.train()
    x, ids, mask, labels, _ = case.inputs()
    with torch.no_grad():
        e = enc(to_dev(x)).permute(1, 0, 2)
        losses = []
        for s in (11, 11, 12, 13, 14, 15):
            dec._seed = torch.tensor([s], dtype=torch.int64, device='cuda')
            losses.append(dec(ids.cuda(), attention_mask=mask.cuda(), encoder_hidden_states=e, labels=labels.cuda()).loss.sum().item())
        dec.eval()
        ref = dec(ids.cuda(), attention_mask=mask.cuda(), encoder_hidden_states=e, labels=labels.cuda()).loss.sum().item()
    assert abs(losses[0] - losses[1]) < 1e-5 * losses[0]      # same masks; fp32 atomics make the last bits order-dependent
    assert len(set(losses[1:])) == 5
    assert abs(np.mean(losses[1:]) - ref) / ref < 0.1


def test_large_vqa_geometry_matches_oracle():
    """BASELINE config 5 geometry (Prismer-LARGE VQA): width 1024 / 16 heads, patch 14 at 480^2 (34x34 rgb tokens), expert maps
    224 -> 256 -> 16x16 tokens (bicubic positional re-grid 34^2 -> 16^2), resampler head dim 128, T = 40 -- with 2 ViT and 2
    decoder layers so the CPU oracle finishes in seconds.  HIP forward + VQA loss + probe gradients vs the oracle."""
    d = config.prismer_large()
    d.vit_layers = 2; d.num_hidden_layers = 2
    enc = VisionTransformer(d.image_resolution, d.patch_size, d.width, d.vit_layers, d.vit_heads, dict(d.experts))
    dec = RobertaForCausalLMModified(_Cfg(d.roberta_config_dict()))
    esd, dsd = synth.synth_encoder_state(d, 5), synth.synth_decoder_state(d, 5)
    enc.load_state_dict(esd); dec.load_state_dict(dsd)
    enc.cuda().eval(); dec.cuda().eval()
    B, T = 1, 40
    x = synth.synth_experts(d, B, seed=9)
    ids, mask, labels = synth.synth_text(d, B, T, seed=9, ragged=False, prompt_length=1)
    labels[:, :35] = -100                                     # prismer_vqa.py:32-33: only the answer span is scored
    weights = torch.tensor([0.7])
    tab = [random.Random(11).randint(0, 127) for _ in range(256)]
    enc.instance_table = torch.tensor(tab, dtype=torch.int32).cuda()
    for p in list(enc.parameters()) + list(dec.parameters()):
        p.requires_grad = False
    probe = ['positional_embedding', 'resampler.latents', 'conv1.depth.13.weight']
    for n in probe:
        dict(enc.named_parameters())[n].requires_grad = True
    e = enc(to_dev(x))
    assert e.shape == (34 * 34 + 64, B, 1024)
    out = dec(ids.cuda(), attention_mask=mask.cuda(), encoder_hidden_states=e.permute(1, 0, 2), labels=labels.cuda(), return_dict=True)
    logits_hip = out.logits.float().cpu()
    (weights.cuda() * out.loss).mean().backward()
    for n in probe:
        esd[n].requires_grad_(True)
    eo = O.vision_encoder(esd, x, d.patch_size, d.vit_heads, False, tab)
    lg, ls = O.text_decoder(dsd, ids, mask, eo.transpose(0, 1), d.num_attention_heads, labels)
    (weights * ls).mean().backward()
    assert rel_fro(e.float(), eo) < TOL_ACT
    assert rel_fro(logits_hip, lg) < TOL_ACT
    assert rel_fro(out.loss, ls) < TOL_LOSS
    for n in probe:
        assert rel_fro(dict(enc.named_parameters())[n].grad, esd[n].grad) < 2 * TOL_GRAD, n


def test_heads_generate_rank_and_vqa_run():
    """next-row smoke: caption generate (beam 3) / rank and the VQA training loss run end-to-end on the HIP
    forward with pre-tokenised ids (no vocabulary on disk), on the tiny geometry."""
    from prismer_amd.model.prismer_caption import PrismerCaption
    from prismer_amd.model.prismer_vqa import PrismerVQA
    case = C.Case('tiny_vqa')
    d = case.dims
    x, ids, mask, labels, weights = case.inputs()

    def make(cls):
        m = cls.__new__(cls)
        torch.nn.Module.__init__(m)
        m.tokenizer = None
        m.expert_encoder, m.text_decoder, _, _ = build(case)
        m.expert_encoder.eval(); m.text_decoder.eval()
        return m
    cap = make(PrismerCaption)
    xs = to_dev(x)
    B = ids.shape[0]
    prefix = (torch.tensor([[0, 83, 2170, 9, 2]] * B), torch.ones(B, 5, dtype=torch.long))          # "<s> A picture of </s>"
    outs = cap(xs, train=False, prefix=prefix, inference='generate')
    assert len(outs) == B and all(4 <= len(o) <= 20 for o in outs)
    answers = (torch.randint(3, d.vocab_size, (6, 3)), torch.ones(6, 3, dtype=torch.long))
    best = cap(xs, answer=answers, train=False, prefix=prefix, inference='rank', k_test=4)
    assert best.shape == (B,) and int(best.max()) < 6
    vqa = make(PrismerVQA)
    vqa.expert_encoder.train()
    q = (ids[:, :8], mask[:, :8]); a = (ids[:, 8:], mask[:, 8:])
    loss = vqa(xs, q, a, weights=weights.cuda(), train=True)
    loss.backward()
    assert torch.isfinite(loss) and vqa.text_decoder.lm_head.dense.weight.grad is not None


@pytest.mark.parametrize('name,head', [('tiny_vqa', 'caption'), ('tiny_vqa', 'vqa'), ('base_b8', 'caption'), ('base_b8', 'vqa')])
def test_rank_inference_matches_oracle(name, head):
    """`inference='rank'` (model/prismer_caption.py:59-112, model/prismer_vqa.py:64-113) on the HIP forward against the oracle's
    restatement (O.rank_answers) at the tiny and the Prismer-BASE geometry: the top-k candidate set picked from the first-token
    probabilities, the length-normalised log-probability of every candidate, and the chosen answer.  bf16 logits can swap
    near-ties, so identity is required only where the oracle's own margin is larger than the bf16 noise."""
    from prismer_amd.model.prismer_caption import PrismerCaption
    from prismer_amd.model.prismer_vqa import PrismerVQA
    case = C.Case(name)
    d = case.dims
    x, ids, mask, _, _ = case.inputs()
    B = min(ids.shape[0], 4)
    x = {k: ({kk: vv[:B] for kk, vv in v.items()} if isinstance(v, dict) else v[:B]) for k, v in x.items()}
    cls = PrismerCaption if head == 'caption' else PrismerVQA
    m = cls.__new__(cls)
    torch.nn.Module.__init__(m)
    m.tokenizer = None
    m.expert_encoder, m.text_decoder, esd, dsd = build(case)
    m.expert_encoder.eval(); m.text_decoder.eval()
    tab = case.instance_table(x)
    m.expert_encoder.instance_table = None if tab is None else torch.tensor(tab, dtype=torch.int32).cuda()
    g = torch.Generator().manual_seed(17)
    n_ans, Ta, k = 12, 4, 5
    a_ids = torch.randint(3, d.vocab_size, (n_ans, Ta), generator=g)
    a_att = torch.ones(n_ans, Ta, dtype=torch.long)
    for i in range(n_ans):                                              # ragged answers ending in </s>, padded to the longest
        L = 2 + i % (Ta - 1)
        a_ids[i, L - 1] = d.eos_token_id
        a_ids[i, L:] = d.pad_token_id; a_att[i, L:] = 0
    if head == 'caption':
        prefix = (torch.tensor([[0, 83 % d.vocab_size, 2170 % d.vocab_size, 9, 2]] * B), torch.ones(B, 5, dtype=torch.long))
        start_ids, start_att = prefix[0][:, :-1], prefix[1][:, :-1]     # prismer_caption.py:70-71: the </s> is dropped
        got = m(to_dev(x), answer=(a_ids, a_att), train=False, prefix=prefix, inference='rank', k_test=k, return_scores=True)
    else:
        Tq = 7
        start_ids = torch.randint(3, d.vocab_size, (B, Tq), generator=g); start_ids[:, 0] = d.bos_token_id
        start_att = torch.ones(B, Tq, dtype=torch.long)
        got = m(to_dev(x), (start_ids, start_att), answer=(a_ids, a_att), train=False, inference='rank', k_test=k, return_scores=True)
    best, topk, lp = [t.cpu() for t in got]
    with torch.no_grad():
        eo = O.vision_encoder(esd, x, d.patch_size, d.vit_heads, False, tab)
        want_best, want_topk, want_lp = O.rank_answers(dsd, eo.transpose(0, 1), start_ids, start_att, a_ids, a_att, k, d.num_attention_heads,
                                                       pad=d.pad_token_id)
    print(name, head, 'best', best.tolist(), want_best.tolist(), 'topk', topk.tolist(), want_topk.tolist())
    agree = 0
    for b in range(B):
        score = {int(i): float(s) for i, s in zip(topk[b], lp[b])}
        wscore = {int(i): float(s) for i, s in zip(want_topk[b], want_lp[b])}
        common = sorted(set(score) & set(wscore))
        assert len(common) >= k - 1, (b, topk[b].tolist(), want_topk[b].tolist())
        for i in common:                                                # length-normalised log-probs: bf16 vs fp32 decoder
            assert abs(score[i] - wscore[i]) < 2e-2 * abs(wscore[i]) + 2e-2, (b, i, score[i], wscore[i])
        srt = sorted(wscore.values(), reverse=True)
        margin = srt[0] - srt[1]
        if margin > 5e-2:
            assert int(best[b]) == int(want_best[b]), (b, score, wscore)
        agree += int(best[b]) == int(want_best[b])
    assert agree >= B - 1


# ---------------------------------------------------------------------------------------------------------------------------
# The path bench.py times: Trainer.step() under hipGraph replay (side streams, deferred grouped weight gradients, merged
# cross-attention K/V projection, fused AdamW beside the encoder backward), at the BASELINE geometries, against outputs of the
# REFERENCE classes (fixtures minted by tests/golden/make_golden.py).  Dropout 0 and a pinned instance table make it deterministic.
class _Holder(torch.nn.Module):
    pass


def _pinned_trainer(case, use_graph, lr=1e-3, p_drop=0.0, **kw):
    from prismer_amd.trainer import Trainer
    enc, dec, _, _ = build(case, p_drop=p_drop)
    set_freeze(enc, dec)
    m = _Holder(); m.expert_encoder, m.text_decoder = enc, dec
    x, ids, mask, labels, weights = case.inputs()
    tab = case.instance_table(x)
    kw.setdefault('keep_grads', True)
    tr = Trainer(m, lr=lr, weight_decay=0.05, total_steps=10, use_graph=use_graph, **kw)
    tr.set_batch(to_dev(x), ids, mask, labels, None if weights is None else weights.cuda())
    if tab is not None:
        orig = tr._host_prologue

        def prologue():
            orig()
            tr.table.copy_(torch.tensor(tab, dtype=torch.int32))
        tr._host_prologue = prologue
    return tr, m


def _check_grads_against_golden(g, named_grads, tag):
    trainable = str(g['requires_grad']).split('\n')
    assert sorted(named_grads) == sorted(trainable)
    worst, worst_p = [], []
    for n in trainable:
        gn = float(g['gnorm.' + n])
        if gn < 1e-4:
            continue
        gr = named_grads[n]
        e_norm = abs(gr.double().norm().item() - gn) / gn
        idx = C.sample_idx(n, gr.numel())
        samp = gr.flatten()[idx.to(gr.device)].float().cpu().numpy()
        e_samp = float(np.linalg.norm(samp - g['gsamp.' + n]) / (np.linalg.norm(g['gsamp.' + n]) + 1e-30))
        bar_s = max(2 * TOL_GRAD_SAMPLED, 2.0 * float(g['ac_samp.' + n]))
        bar_n = max(TOL_GRAD, 2.0 * float(g['ac_norm.' + n]))
        # A 16-entry sample of a SPARSE gradient (the instance-embedding table: most rows are never drawn, vit.py:141-148; position rows beyond
        # the text length; dead squared-ReLU units) can hold fewer than six non-zero entries: its relative error is then an estimate from a
        # handful of numbers of magnitude 1e-6 (round 5, large_vqa_b4 instance_embedding, 3 non-zero samples: 0.035 with one LayerNorm forward
        # kernel, 0.142 with another of IDENTICAL accuracy against fp64 -- while the full-tensor error of that gradient is 0.016 / 0.021, the
        # autocast yardstick's own 0.020).  Such tensors are held to the full-tensor projection check below and to the norm, not to the sample.
        if 'gproj.' + n in g and int((g['gsamp.' + n] != 0).sum()) < 6:
            e_samp = 0.0
        worst.append((e_samp / bar_s, e_norm / bar_n, e_samp, e_norm, n))
        if 'gfull.' + n in g:
            # full-tensor comparison: the yardstick is the FULL-tensor error of PyTorch's own autocast (ac_rel) where the fixture has it -- the 16-entry
            # sample estimate ac_samp under-reads it (large_vqa_b4, conv1.depth.2.weight: 0.057 sampled, 0.109 over the whole tensor; this path 0.118);
            # same multipliers as the projection check below (stems 1.5x, everything else 2x)
            bar_full = max(TOL_GRAD, (1.5 if '.conv1.' in n else 2.0) * float(g['ac_rel.' + n])) if 'ac_rel.' + n in g else max(bar_s, TOL_GRAD)
            assert rel_fro(gr, torch.from_numpy(g['gfull.' + n])) < bar_full, (n, bar_full)
        if 'gproj.' + n in g:
            # FULL-tensor check (round 3): N_PROJ fixed random projections of the whole gradient estimate |g_hip - g_ref|_F; the
            # bar is the exact full-tensor error of PyTorch's own bf16 autocast on the reference modules (ac_rel), with the
            # 1.8x head-room a 16-projection chi-square estimate needs at the 1e-5 level
            e_proj = C.projected_error(C.grad_projections(n, gr), g['gproj.' + n]) / gn
            # (stems: 1.5x autocast since round 4 -- their full-tensor error sits at 0.95x autocast's in the median, 1.42x at worst, at bs32)
            bar_p = PROJ_SLACK * max(TOL_GRAD, (1.5 if '.conv1.' in n else 2.0) * float(g['ac_rel.' + n]))
            worst_p.append((e_proj / bar_p, e_proj, float(g['ac_rel.' + n]), n))
    worst.sort(reverse=True)
    med = float(np.median([w[2] for w in worst]))
    print(tag, 'worst (samp/bar, norm/bar, samp, norm, name):', worst[:3], 'median sampled error', med)
    assert worst[0][0] < 1.0 and max(w[1] for w in worst) < 1.0, worst[:4]
    assert med < 3e-2
    if worst_p:
        worst_p.sort(reverse=True)
        med_p = float(np.median([w[1] for w in worst_p]))
        print(tag, 'projections: worst (err/bar, full-tensor error estimate, autocast full-tensor error, name):', worst_p[:3], 'median', med_p)
        stem = sorted(((w[1] / max(w[2], 1e-9), w[1], w[2], w[3]) for w in worst_p if '.conv1.' in w[3] and w[2] > 0), reverse=True)
        if stem:
            print(tag, 'stems: worst (error / autocast error, error, autocast error, name):', stem[:4], 'median ratio', float(np.median([x[0] for x in stem])))
        assert worst_p[0][0] < 1.0, worst_p[:4]
        assert med_p < 3e-2


PROJ_SLACK = 1.8


@pytest.mark.parametrize('name', ['base_b8', 'zbase_b4', 'large_vqa_b1', 'base_b32', 'large_vqa_b4', 'huge_b1', 'zbase_b32', 'large_vqa_b16'])      # ('large_vqa_b1+big', the forced-dispatch
# variant of rounds 3-5, is superseded by large_vqa_b4 / large_vqa_b16, where the dispatch picks the 256x128 kernels unforced; `name + '+big'` still works)
def test_trainer_hipgraph_step_matches_reference_golden(name):
    force_big = name.endswith('+big')           # LARGE shapes (H = 1024, 24 + 24 layers, S = 1220) through the 256x128 ping-pong kernel:
    name = name.split('+')[0]                   # at B = 1 the cost model never picks it, config 5's bs16 does
    from prismer_amd import _lib as _l
    if force_big:
        _l.lib.ph_gemm_tuning(_l.GEMM_BIG_DEFAULT[0], 1)
    try:
        _hipgraph_step_vs_golden(name, force_big)
    finally:
        _l.lib.ph_gemm_tuning(*_l.GEMM_BIG_DEFAULT)


def _hipgraph_step_vs_golden(name, force_big=False):
    """Prismer-BASE B=8 (BASELINE config 3 geometry), PrismerZ-BASE B=4 (config 2), Prismer-LARGE VQA 480^2 B=1 (config 5): full
    depth.  First replayed step: loss, every trainable gradient (norm + sampled entries, autocast yardstick), BatchNorm
    running statistics after exactly ONE update, and the fused AdamW result."""
    from prismer_amd.trainer import cosine_lr
    import ctypes
    from prismer_amd import _lib
    g = np.load(os.path.join(GOLD, name + '.npz'))
    case = C.Case(name)
    lr = 1e-4
    counts = (ctypes.c_int64 * 16)()
    _lib.lib.ph_gemm_dispatch_counts(counts, 16, 1)             # reset: what the capture of this Trainer launches is counted below
    tr, m = _pinned_trainer(case, use_graph=True, lr=lr)
    p0 = [st.master[:st.n_train].clone() for st in tr.stores]
    loss = tr.step()
    torch.cuda.synchronize()
    assert tr.use_graph and tr.graphs is not None and tr.it == 1
    ncls = _lib.lib.ph_gemm_dispatch_counts(counts, 16, 0)
    by_class = dict(zip(('128x128', '64x64', 'ks2', 'big', 'big_grouped', 'grouped', 'splitk_reduce'), list(counts)[:ncls]))
    print(name, 'GEMM launches by kernel class during warm-up + capture:', by_class)
    if force_big:
        assert by_class['big'] > 0, by_class
    if name in ('zbase_b32', 'large_vqa_b16'):
        # round 6: the secondary legs of the bench line at THEIR benchmarked batch (config 2 at bs32: M = 32 x 196; config 5 at bs16: M = 16 x 1220) --
        # the dispatch (tile choice, tail split, grouped forms) depends on M, so the kernels the bench runs are the kernels this fixture pins
        assert by_class['big'] > 0 and by_class['big_grouped'] > 0 and by_class['ks2'] > 0, by_class
    if name == 'large_vqa_b4':
        # config 5 at a batch where the dispatch itself (no ph_gemm_tuning override) puts the LARGE shapes -- M = 4 x 1220 rows, K = 1024 /
        # 4096, ragged text -- on the 256x128 ping-pong kernel and its grouped weight-gradient form (round-3 review: B = 1 had to force them)
        assert by_class['big'] > 0 and by_class['big_grouped'] > 0, by_class
    if name == 'base_b32':
        # the benchmark configuration itself: the 256x128 ping-pong kernel (M = 8320: N = 768 launches and long reductions) and its
        # grouped persistent form (long-reduction weight gradients) must be ON the path this fixture pins, not just unit-tested
        assert by_class['big'] > 0 and by_class['big_grouped'] > 0 and by_class['ks2'] > 0 and by_class['128x128'] > 0, by_class
    assert math_close(loss.item(), float(g['total_train']), TOL_LOSS), (loss.item(), float(g['total_train']))
    named = {}
    for pref, st in (('expert_encoder.', tr.stores[0]), ('text_decoder.', tr.stores[1])):
        for nm in st.names:
            if st.is_trainable(nm):
                named[pref + nm] = st.g(nm).detach()
    _check_grads_against_golden(g, named, name)
    for k, v in m.expert_encoder.state_dict().items():          # exactly one running-stat update (capture warm-up is side-effect free)
        if 'running_' in k:
            assert rel_fro(v, torch.from_numpy(g['bn.' + k])) < 5e-3, k
        if 'num_batches' in k:
            assert int(v) == int(g['bn.' + k]) == 1
    # fused AdamW, first step: p <- p (1 - lr wd) - lr g / (|g| + eps)   (train_caption.py:111-112,127: lr = cosine(0) = init_lr)
    assert cosine_lr(0, 10, lr, 0.0) == lr
    for st, before in zip(tr.stores, p0):
        gr = st.grad[:st.n_train]
        want = before * (1.0 - lr * 0.05) - lr * gr / (gr.abs() + 1e-8)
        err = (st.master[:st.n_train] - want).abs().max().item()
        assert err < 2e-6, err
    # ... and against the REFERENCE gradients: wherever the reference gradient is not round-off, the update has its sign
    agree = total = 0
    for pref, st, before in (('expert_encoder.', tr.stores[0], p0[0]), ('text_decoder.', tr.stores[1], p0[1])):
        for nm in st.names:
            if not st.is_trainable(nm) or float(g['gnorm.' + pref + nm]) < 1e-4:
                continue
            idx = C.sample_idx(pref + nm, st.numel[nm]).cuda()
            o = st.offset[nm]
            delta = (st.master[o:o + st.numel[nm]][idx] - before[o:o + st.numel[nm]][idx] * (1.0 - lr * 0.05)).cpu().numpy()
            ref = g['gsamp.' + pref + nm]
            big = np.abs(ref) > 0.05 * np.abs(ref).max() + 1e-7
            agree += int((np.sign(delta[big]) == -np.sign(ref[big])).sum()); total += int(big.sum())
    assert agree >= 0.97 * total, (agree, total)
    loss2 = tr.step()
    torch.cuda.synchronize()
    assert np.isfinite(loss2.item()) and tr.it == 2


@pytest.mark.parametrize('name', list(C.DROP_CASES))
def test_trainer_hipgraph_dropout_steps_match_reference_golden(name):
    """Deterministic REFERENCE parity at dropout 0.1 -- the arithmetic every benchmarked step runs (configs/prismer.json:4,8;
    roberta.py:75,123,138,181).  tests/golden/<case>_drop.npz holds the reference classes' loss and gradients in full training mode with
    every nn.Dropout replaced by the masks this library draws for a given device seed (pure functions of (seed, call site, element):
    tests/util.LibraryDropout, pinned to Random123 vectors in test_philox_kat_cpu.py; minted by tests/golden/make_golden.py <case>+drop).
    Here: the native Trainer under hipGraph REPLAY, learning rate 0 so that step 2 sees the same weights with the NEXT seed (the device
    seed is advanced by a kernel inside the graph: fresh masks per replay).  Same bars as the dropout-free test.  Pins the Philox-7
    generator, the embedding / GEMM-epilogue / LayerNorm-backward mask regeneration and the quad-cooperative dK/dV dropout words end to end."""
    from tests.util import splitmix64
    g = np.load(os.path.join(GOLD, name + '_drop.npz'))
    case = C.Case(name)
    tr, m = _pinned_trainer(case, use_graph=True, lr=0.0, p_drop=0.1)
    assert case.dims.hidden_dropout_prob == case.dims.attention_probs_dropout_prob == 0.1
    seeds = [int(v) for v in g['seeds']]
    assert seeds == [C.DROP_SEED, splitmix64(C.DROP_SEED)]
    tr.seed.copy_(torch.tensor([C.DROP_SEED - (1 << 64) if C.DROP_SEED >= (1 << 63) else C.DROP_SEED], dtype=torch.int64))
    p0 = [st.master.clone() for st in tr.stores]
    for si in (1, 2):
        assert int(tr.seed.item()) & 0xFFFFFFFFFFFFFFFF == seeds[si - 1]
        loss = tr.step()
        torch.cuda.synchronize()
        assert tr.use_graph and tr.graphs is not None and tr.it == si
        want = float(g[f's{si}.total_train'])
        print(name, 'step', si, 'loss', loss.item(), 'reference under the same masks', want)
        assert math_close(loss.item(), want, TOL_LOSS), (si, loss.item(), want)
        gv = {k[3:]: g[k] for k in g.files if k.startswith(f's{si}.')}
        gv['requires_grad'] = g['requires_grad']
        named = {}
        for pref, st in (('expert_encoder.', tr.stores[0]), ('text_decoder.', tr.stores[1])):
            for nm in st.names:
                if st.is_trainable(nm):
                    named[pref + nm] = st.g(nm).detach()
        _check_grads_against_golden(gv, named, f'{name} dropout step {si}')
    for st, before in zip(tr.stores, p0):                     # lr = 0: the weights did not move (step 2 is the same model, other masks)
        assert torch.equal(st.master, before)
    assert abs(float(g['s1.total_train']) - float(g['s2.total_train'])) > 1e-4 * float(g['s1.total_train'])


@pytest.mark.parametrize('name', list(C.TRAJ_CASES))
def test_trainer_hipgraph_trajectory_matches_reference_loop(name):
    """Steps 2+ against the REFERENCE (round 6): tests/golden/<case>_traj.npz holds C.TRAJ_STEPS iterations of the reference training loop
    (train_caption.py:111-112 torch.optim.AdamW, :126-135 cosine_lr_schedule -> forward -> zero_grad -> backward -> step) on the reference classes in
    full training mode (BatchNorm batch statistics, dropout 0.1 under this library's masks, the seed advanced per step exactly as the device
    seed is), minted by tests/golden/make_golden.py <case>+traj, plus the same loop under PyTorch's own bf16 autocast as the yardstick.  Here:
    the native Trainer under hipGraph REPLAY walks the same four steps -- loss curve, parameter displacement p_4 - p_0 of every trainable tensor
    (16 fixed projections: a full-tensor estimate) and the BatchNorm running statistics after four updates.  AdamW's first steps move every entry
    by ~lr whatever the gradient's size, so entries whose gradient is round-off flip sign under ANY bf16 arithmetic: the yardstick's own
    displacement error is 8 % in the median; the bar is 2x the yardstick per tensor (floor 15 %), the loss curve is held to 2e-3."""
    g = np.load(os.path.join(GOLD, name + '_traj.npz'))
    case = C.Case(name)
    assert (C.TRAJ_LR, C.TRAJ_MIN_LR, C.TRAJ_WD, C.TRAJ_TOTAL) == (5e-5, 0.0, 0.05, 10)         # what _pinned_trainer builds (configs/caption.yaml:12-14)
    tr, m = _pinned_trainer(case, use_graph=True, lr=C.TRAJ_LR, p_drop=0.1, keep_grads=False)
    tr.seed.copy_(torch.tensor([C.DROP_SEED - (1 << 64) if C.DROP_SEED >= (1 << 63) else C.DROP_SEED], dtype=torch.int64))
    p0 = [st.master.clone() for st in tr.stores]
    losses = []
    for k in range(C.TRAJ_STEPS):
        losses.append(tr.step().item())
    torch.cuda.synchronize()
    assert tr.use_graph and tr.graphs is not None and tr.it == C.TRAJ_STEPS
    ref, ac = g['losses'], g['ac_losses']
    print(name, 'loss curve', losses, 'reference', ref.tolist(), 'autocast yardstick', ac.tolist())
    for k in range(C.TRAJ_STEPS):
        bar = max(TOL_LOSS, 3.0 * abs(ac[k] - ref[k]) / ref[k])
        assert math_close(losses[k], float(ref[k]), bar), (k, losses[k], float(ref[k]))
    assert ref[-1] < 0.9 * ref[0]                                       # the fixture itself trains
    trainable = str(g['requires_grad']).split('\n')
    rows = []
    for pref, st, before in (('expert_encoder.', tr.stores[0], p0[0]), ('text_decoder.', tr.stores[1], p0[1])):
        for nm in st.names:
            if not st.is_trainable(nm):
                continue
            n = pref + nm
            assert n in trainable
            dn = float(g['dnorm.' + n])
            if dn < 1e-6:
                continue
            o = st.offset[nm]
            delta = st.master[o:o + st.numel[nm]] - before[o:o + st.numel[nm]]
            e = C.projected_error(C.grad_projections(n, delta), g['dproj.' + n]) / dn
            bar = PROJ_SLACK * max(0.15, 2.0 * float(g['ac_drel.' + n]))
            rows.append((e / bar, e, float(g['ac_drel.' + n]), abs(delta.double().norm().item() - dn) / dn, n))
    rows.sort(reverse=True)
    med = float(np.median([r[1] for r in rows]))
    print(name, 'displacement after', C.TRAJ_STEPS, 'steps: worst (err/bar, full-tensor error estimate, autocast error, norm error, name)', rows[:3], 'median', med,
          'autocast median', float(np.median([r[2] for r in rows])))
    assert rows[0][0] < 1.0, rows[:4]
    # norm of the displacement: same yardstick (a tensor whose fp32 gradient is EXACTLY zero in most entries -- untouched embedding rows: only the
    # weight decay moves them -- picks up +-lr from round-off gradients under any bf16 arithmetic, PyTorch's autocast included)
    worst_n = max(rows, key=lambda r: r[3] / max(0.1, 2.0 * r[2]))
    print(name, 'worst displacement norm (norm error, autocast full-tensor error, name):', worst_n[3], worst_n[2], worst_n[4])
    assert med < 0.15 and worst_n[3] < max(0.1, 2.0 * worst_n[2]), worst_n
    for k, v in m.expert_encoder.state_dict().items():
        if 'running_' in k:
            assert rel_fro(v, torch.from_numpy(g['bn.' + k])) < 1e-2, k
        if 'num_batches' in k:
            assert int(v) == int(g['bn.' + k]) == C.TRAJ_STEPS


def test_bench_gradient_handling_equals_the_pinned_one():
    """bench.py runs keep_grads=False (AdamW zeroes the gradients except the single-writer ones, which their GEMM overwrites); the
    reference-pinned tests above read gradients, i.e. run keep_grads=True (everything accumulates into a buffer filled at the start
    of the step).  Same batch, same weights, three steps under hipGraph replay at Prismer-BASE B = 8: the two must walk the same
    trajectory -- losses and parameters to the atomics noise of two runs."""
    case = C.Case('base_b8')
    res = []
    for keep in (True, True, False):                            # the repeated run measures the run-to-run noise (fp32 atomics order, +-lr moves)
        tr, m = _pinned_trainer(case, use_graph=True, lr=1e-4, keep_grads=keep)
        losses = [tr.step().item() for _ in range(3)]
        torch.cuda.synchronize()
        assert tr._exclusive is not None and len(tr._exclusive) > 50
        res.append((losses, [st.master[:st.n_train].clone() for st in tr.stores], [t.clone() for t in tr.m]))
        del tr, m
    (la, pa, ma), (ln, pn, mn), (lb, pb, mb) = res
    assert math_close(la[0], lb[0], 1e-5), (la, lb)
    assert math_close(la[1], lb[1], 1e-3) and math_close(la[2], lb[2], 2e-3), (la, lb)
    for ta, tn, tb in zip(pa + ma, pn + mn, pb + mb):
        noise = rel_fro(tn, ta)
        assert rel_fro(tb, ta) < max(2.0 * noise, 1e-3), (rel_fro(tb, ta), noise)


@pytest.mark.parametrize('keep_grads', [True, False])
def test_trainer_graph_equals_eager_first_step(keep_grads):
    """(keep_grads=False is the bench's setting: the fused AdamW zeroes the gradients and the captured forward carries no fill.)
    capture warm-up must not leak into the training state: the first replayed step and the first eager step produce the same
    loss, gradients, parameters, Adam moments, BatchNorm buffers and iteration counter (tiny geometry, dropout ON: the
    dropout seed is part of the state)."""
    case = C.Case('tiny_caption')
    res = []
    for use_graph in (False, True):
        from prismer_amd.trainer import Trainer
        enc, dec, _, _ = build(case)                            # hidden / attention dropout 0.1
        set_freeze(enc, dec)
        dec._seed = torch.tensor([1234567], dtype=torch.int64, device='cuda')
        m = _Holder(); m.expert_encoder, m.text_decoder = enc, dec
        x, ids, mask, labels, _ = case.inputs()
        tr = Trainer(m, lr=1e-3, total_steps=10, use_graph=use_graph, keep_grads=keep_grads)
        tr.set_batch(to_dev(x), ids, mask, labels)
        random.seed(99)
        losses = [tr.step().item() for _ in range(2)]
        if not keep_grads:
            # left clean for the next step: everything except the single-writer weight gradients, which the next step's GEMM overwrites
            assert len(tr._exclusive) > 10
            for st, rs in zip(tr.stores, tr._excl_ranges):
                ranges = []                                       # union of the overwritten outputs (q / kv parts of one in_proj_weight touch)
                for a, k in sorted(rs):
                    if ranges and ranges[-1][0] + ranges[-1][1] == a:
                        ranges[-1] = (ranges[-1][0], ranges[-1][1] + k)
                    else:
                        ranges.append((a, k))
                for nm in st.names:
                    if not st.is_trainable(nm):
                        continue
                    o, n = st.offset[nm], st.numel[nm]
                    if not any(a <= o and o + n <= a + k for a, k in ranges):
                        assert float(st.g(nm).abs().max()) == 0.0, nm
        torch.cuda.synchronize()
        res.append(dict(losses=losses, it=tr.it, seed=int(tr.seed.item()), p=[st.master.clone() for st in tr.stores],
                        m=[t.clone() for t in tr.m], bufs=[b.clone().float() for b in enc.buffers()]))
    a, b = res
    assert a['it'] == b['it'] == 2 and a['seed'] == b['seed']
    assert math_close(a['losses'][0], b['losses'][0], 1e-5), (a['losses'], b['losses'])     # same state, same masks, same table
    # step 2 sees parameters after one AdamW step: near-zero gradients (atomics-order noise) can take either sign of the +-lr move
    assert math_close(a['losses'][1], b['losses'][1], 1e-3), (a['losses'], b['losses'])
    for ta, tb in zip(a['p'] + a['m'] + a['bufs'], b['p'] + b['m'] + b['bufs']):
        assert rel_fro(tb, ta) < 5e-3                           # same kernels; fp32 atomics order differs between runs


def test_trainer_resume_from_state_dict():
    """Trainer.state_dict / load_state_dict (the content of accelerate's save_state, train_caption.py:174: model + optimizer + RNG):
    a second Trainer that has already diverged (one step on its own), loaded with the state after two steps of the first, takes the
    same third step -- loss, parameters, Adam moments, BatchNorm buffers, iteration counter, dropout seed."""
    from prismer_amd.trainer import Trainer
    case = C.Case('tiny_caption')
    x, ids, mask, labels, _ = case.inputs()

    def make():
        enc, dec, _, _ = build(case)
        set_freeze(enc, dec)
        dec._seed = torch.tensor([1234567], dtype=torch.int64, device='cuda')
        m = _Holder(); m.expert_encoder, m.text_decoder = enc, dec
        tr = Trainer(m, lr=1e-3, total_steps=10, use_graph=True)
        tr.set_batch(to_dev(x), ids, mask, labels)
        return tr, enc

    tr_a, enc_a = make()
    random.seed(99)
    for _ in range(2):
        tr_a.step()
    torch.cuda.synchronize()
    sd = tr_a.state_dict()
    loss_a = tr_a.step().item()
    tr_b, enc_b = make()
    random.seed(5)
    tr_b.step()                                                  # its own first step: every piece of state now differs from sd
    tr_b.load_state_dict(sd)
    assert tr_b.it == 2 and int(tr_b.seed.item()) == int(sd['seed'].item())
    loss_b = tr_b.step().item()
    torch.cuda.synchronize()
    assert tr_a.it == tr_b.it == 3 and int(tr_a.seed.item()) == int(tr_b.seed.item())
    assert math_close(loss_a, loss_b, 1e-3), (loss_a, loss_b)
    for ta, tb in zip([st.master for st in tr_a.stores] + tr_a.m + tr_a.v + [b.float() for b in enc_a.buffers()],
                      [st.master for st in tr_b.stores] + tr_b.m + tr_b.v + [b.float() for b in enc_b.buffers()]):
        assert rel_fro(tb, ta) < 5e-3
    with pytest.raises(ValueError):
        bad = dict(sd); bad['m'] = sd['m'][:-1] if len(sd['m']) > 1 else [sd['m'][0][:-1]]
        tr_b.load_state_dict(bad)


def test_set_batch_pads_text_and_rejects_other_shapes():
    case = C.Case('tiny_caption')
    tr, _ = _pinned_trainer(case, use_graph=False, max_text_len=16)
    x, ids, mask, labels, _ = case.inputs()
    assert tr.static['input_ids'].shape == (2, 16)
    l_pad = tr.step().item()
    tr2, _ = _pinned_trainer(case, use_graph=False)
    l_ref = tr2.step().item()
    assert math_close(l_pad, l_ref, 1e-4), (l_pad, l_ref)       # pads are masked keys / ignored labels: same loss
    tr.set_batch(to_dev(x), ids[:, :10], mask[:, :10], labels[:, :10])            # shorter batch: padded into the static buffers
    assert int(tr.static['attention_mask'][:, 10:].sum()) == 0 and int((tr.static['labels'][:, 10:] != -100).sum()) == 0
    with pytest.raises(ValueError):
        tr.set_batch(to_dev(x), ids[:1], mask[:1], labels[:1])
    with pytest.raises(ValueError):
        tr.set_batch(to_dev(x), ids, mask, labels, torch.ones(2).cuda())


def test_prefetched_batches_equal_bound_batches():
    """Trainer.prefetch_batch + commit_prefetched (host batch i+1 staged on a copy stream while step i runs, moved into the static
    buffers device-to-device) leaves exactly the inputs set_batch would have bound, for pinned host tensors, shorter captions and
    repeated use; the loss of the following step is the loss of the bound batch."""
    case = C.Case('tiny_caption')
    tr, _ = _pinned_trainer(case, use_graph=False, max_text_len=16)
    x, ids, mask, labels, _ = case.inputs()
    xd = to_dev(x)

    def host(t):
        return {k: host(v) for k, v in t.items()} if isinstance(t, dict) else t.cpu().pin_memory()

    def leaves(t):
        if isinstance(t, dict):
            for k in sorted(t):
                yield from leaves(t[k])
        elif t is not None:
            yield t
    x2 = {k: (v.flip(0) if not isinstance(v, dict) else {kk: vv.flip(0) for kk, vv in v.items()}) for k, v in xd.items()}     # another batch: rows swapped
    want = []
    for bx, n in ((x2, 10), (xd, 16), (x2, 12)):
        tr.set_batch(bx, ids[:, :n], mask[:, :n], labels[:, :n])
        torch.cuda.synchronize()
        want.append(([t.clone() for t in leaves(tr.static)], tr.step().item()))
    tr2, _ = _pinned_trainer(case, use_graph=False, max_text_len=16)
    with pytest.raises(RuntimeError):
        type(tr2).prefetch_batch(type('T', (), {'static': None})(), None, None, None, None)
    tr2.prefetch_batch(host(x2), ids[:, :10].cpu().pin_memory(), mask[:, :10].cpu().pin_memory(), labels[:, :10].cpu().pin_memory())
    for i, (bx, n) in enumerate(((xd, 16), (x2, 12), (xd, 16))):
        tr2.commit_prefetched()
        tr2.prefetch_batch(host(bx), ids[:, :n].cpu().pin_memory(), mask[:, :n].cpu().pin_memory(), labels[:, :n].cpu().pin_memory())      # travels while the step runs
        torch.cuda.synchronize()
        for a, b in zip(leaves(tr2.static), want[i][0]):
            assert torch.equal(a, b)
        loss = tr2.step().item()
        assert math_close(loss, want[i][1], 2e-3), (i, loss, want[i][1])


def test_prefetch_api_misuse_raises():
    case = C.Case('tiny_caption')
    tr, _ = _pinned_trainer(case, use_graph=False, max_text_len=16)
    x, ids, mask, labels, _ = case.inputs()
    with pytest.raises(RuntimeError, match='nothing is staged'):
        tr.commit_prefetched()
    tr.prefetch_batch(to_dev(x), ids, mask, labels)                     # device-resident sources: ordered behind their producers
    with pytest.raises(RuntimeError, match='waiting for commit'):
        tr.prefetch_batch(to_dev(x), ids, mask, labels)
    tr.commit_prefetched()
    with pytest.raises(RuntimeError, match='nothing is staged'):
        tr.commit_prefetched()
    assert np.isfinite(tr.step().item())


def test_loader_fed_graph_replay_without_host_sync_matches_set_batch_trajectory():
    """The scenario of the round-4 defect (bench.py loader leg reported losses of 1e8 .. 1e18): other Trainers have lived and died in the
    process (allocator and runtime state), then a loader-fed Trainer runs under hipGraph REPLAY with prefetch_batch / commit_prefetched and
    NO host synchronisation between commit, prefetch and step.  Checked: the static input buffers after every commit are bit-for-bit the
    host batch (device-side copies taken on the compute stream), and the per-step loss trajectory equals the set_batch-per-step form of
    an identically initialised Trainer to 2e-3 for 10 steps.  Root cause found in round 5: not the staging logic -- the loss accumulator
    of ph_ce_fwd was zeroed by a hipMemsetAsync NODE, which this ROCm replays wrongly (tests/test_kernels_gpu.py::
    test_cross_entropy_under_graph_replay_needs_no_zeroed_buffers, tools/graph_memset_probe.py)."""
    import gc
    import bench
    for wl, bs in (('z_base_caption', 8), ('base_caption', 4)):           # predecessors: build, capture, step, die
        t0, _, _ = bench.build_trainer(bs, True, 0, workload=wl)
        for _ in range(3):
            l0 = t0.step()
        assert np.isfinite(l0.item())
        del t0
        gc.collect(); torch.cuda.empty_cache()
    B, steps = 8, 10

    def leaves(t):
        if isinstance(t, dict):
            for k in sorted(t):
                yield from leaves(t[k])
        elif t is not None:
            yield t

    def pin(t):
        return {k: pin(v) for k, v in t.items()} if isinstance(t, dict) else t.cpu().pin_memory()
    tr, dims, _ = bench.build_trainer(B, True, 0, compact_labels=True)
    batches = []
    for i in range(3):
        x, ids, mask, labels = bench.make_inputs(dims, B, 30, 4321 + i, torch.device('cuda'), True)
        batches.append((pin(x), ids.cpu().pin_memory(), mask.cpu().pin_memory(), labels.cpu().pin_memory()))
    tr.set_batch(*batches[0]); tr.step()
    assert tr.graphs is not None
    tr.prefetch_batch(*batches[1])
    snaps, losses = [], []
    for i in range(steps):                                               # no device synchronisation in here (prefetch_batch waits for the previous COMMIT only)
        tr.commit_prefetched()
        snaps.append([t.clone() for t in leaves({k: v for k, v in tr.static.items() if k != 'dloss'})])
        if i % 2:                                                        # both call orders are legal (the recommended one is step -> prefetch)
            tr.prefetch_batch(*batches[(i + 2) % 3])
            losses.append(tr.step().clone())
        else:
            losses.append(tr.step().clone())
            tr.prefetch_batch(*batches[(i + 2) % 3])
    torch.cuda.synchronize()
    for i, snap in enumerate(snaps):
        b = batches[(i + 1) % 3]
        want = list(leaves(dict(experts=b[0], input_ids=b[1], attention_mask=b[2], labels=b[3])))
        assert len(want) == len(snap)
        for got, w in zip(snap, want):
            assert torch.equal(got.cpu(), w), i
    del tr
    gc.collect(); torch.cuda.empty_cache()
    tr, _, _ = bench.build_trainer(B, True, 0, compact_labels=True)
    tr.set_batch(*batches[0]); tr.step()
    ref = []
    for i in range(steps):
        tr.set_batch(*batches[(i + 1) % 3])
        ref.append(tr.step().clone())
    torch.cuda.synchronize()
    got, ref = [l.item() for l in losses], [l.item() for l in ref]
    print('loader-fed losses', got, 'set_batch losses', ref)
    for a, b in zip(got, ref):
        assert math_close(a, b, 2e-3), (got, ref)
    assert got[-1] < got[0]


def _head(cls, case, train_enc=True):
    m = cls.__new__(cls)
    torch.nn.Module.__init__(m)
    m.tokenizer = None
    m.expert_encoder, m.text_decoder, _, _ = build(case)
    set_freeze(m.expert_encoder, m.text_decoder)
    m.expert_encoder.train(train_enc); m.text_decoder.eval()
    x = case.inputs()[0]
    tab = case.instance_table(x)
    m.expert_encoder.instance_table = None if tab is None else torch.tensor(tab, dtype=torch.int32).cuda()
    return m


@pytest.mark.parametrize('name', ['tiny_caption', 'base_b8'])
def test_caption_head_train_forward_matches_reference_golden(name):
    """PrismerCaption.forward(train=True) (prismer_caption.py:17-34): the head builds the labels itself (pads and the prompt ->
    -100) and returns loss.mean(); compared with the reference's total on the same ids."""
    from prismer_amd.model.prismer_caption import PrismerCaption
    g = np.load(os.path.join(GOLD, name + '.npz'))
    case = C.Case(name)
    cap = _head(PrismerCaption, case)
    x, ids, mask, labels, _ = case.inputs()
    loss = cap(to_dev(x), caption=(ids, mask), prefix=4, train=True)
    assert math_close(loss.item(), float(g['total_train']), TOL_LOSS), (loss.item(), float(g['total_train']))
    loss.backward()
    for n in ('text_decoder.lm_head.dense.bias', 'expert_encoder.resampler.latents'):
        p = dict(cap.named_parameters())[n]
        assert rel_fro(p.grad, torch.from_numpy(g['gfull.' + n])) < max(TOL_GRAD, 4 * float(g['ac_samp.' + n])), n


def test_vqa_head_train_forward_matches_reference_golden():
    """PrismerVQA.forward(train=True) (prismer_vqa.py:18-42): question ‖ answer concatenation with pads in the middle of a row,
    answer-span targets, (weights * loss).mean()."""
    from prismer_amd.model.prismer_vqa import PrismerVQA
    g = np.load(os.path.join(GOLD, 'tiny_vqa_head.npz'))
    case = C.Case('tiny_vqa_head')
    vqa = _head(PrismerVQA, case)
    x = case.inputs()[0]
    q_ids, q_att, a_ids, a_att, weights = case.vqa_parts()
    loss = vqa(to_dev(x), (q_ids, q_att), (a_ids, a_att), weights=weights.cuda(), train=True)
    assert math_close(loss.item(), float(g['total_train']), TOL_LOSS), (loss.item(), float(g['total_train']))
    loss.backward()
    named = dict(vqa.named_parameters())
    grads = {n: named[n].grad for n in str(g['requires_grad']).split('\n')}
    _check_grads_against_golden(g, grads, 'vqa head')


def test_autograd_path_dropout_masks_match_between_forward_and_backward():
    """drop-in path with dropout ON: the backward must regenerate the masks of ITS forward (the persistent seed has already
    moved on by then).  Oracle for the masks = the native Trainer, which runs forward and backward under one seed value."""
    from prismer_amd.trainer import Trainer
    case = C.Case('tiny_caption')
    x, ids, mask, labels, _ = case.inputs()
    tab = case.instance_table(x)
    seed0 = 424242
    # native
    enc, dec, _, _ = build(case)
    set_freeze(enc, dec)
    dec._seed = torch.tensor([seed0], dtype=torch.int64, device='cuda')
    m = _Holder(); m.expert_encoder, m.text_decoder = enc, dec
    tr = Trainer(m, lr=0.0, weight_decay=0.0, total_steps=10, use_graph=False, keep_grads=True)
    tr.set_batch(to_dev(x), ids, mask, labels)
    orig = tr._host_prologue

    def prologue():
        orig()
        tr.table.copy_(torch.tensor(tab, dtype=torch.int32))
    tr._host_prologue = prologue
    l_nat = tr.step().item()
    g_nat = torch.cat([st.grad[:st.n_train].clone() for st in tr.stores])
    # drop-in (autograd)
    enc2, dec2, _, _ = build(case)
    set_freeze(enc2, dec2)
    enc2.train(); dec2.train()
    dec2._seed = torch.tensor([seed0], dtype=torch.int64, device='cuda')
    enc2.instance_table = torch.tensor(tab, dtype=torch.int32).cuda()
    e = enc2(to_dev(x))
    out = dec2(ids.cuda(), attention_mask=mask.cuda(), encoder_hidden_states=e.permute(1, 0, 2), labels=labels.cuda(), return_dict=True)
    assert int(dec2._seed.item()) != seed0                      # the persistent seed advanced right after the forward
    out.loss.mean().backward()
    assert math_close(out.loss.mean().item(), l_nat, 1e-4)
    g_auto = torch.cat([torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten().float()
                                   for n, p in sorted(((n, p) for n, p in mod.named_parameters() if p.requires_grad),
                                                      key=lambda t: mod._store.offset[t[0]])])
                        for mod in (enc2, dec2)])
    g_nat_c = torch.cat([torch.cat([st.g(n).flatten() for n in st.names if st.is_trainable(n)]) for st in tr.stores])
    assert g_auto.shape == g_nat_c.shape
    assert rel_fro(g_auto, g_nat_c) < 2e-3, float(rel_fro(g_auto, g_nat_c))


def test_trainer_pretrain_recipe_freeze_lang_vision_and_warmup():
    """pre-training recipe (train_pretrain.py:100-122, configs/pretrain.yaml:12-21): freeze = 'freeze_lang_vision' (ViT blocks and
    the RoBERTa self-attention / MLP sub-layers frozen; cross-attention, adaptors, stems, resampler, embeddings, LM head train),
    epoch-cosine + linear warm-up schedule, no prompt masking (prefix '').  Gradients vs the CPU oracle's autograd under the same
    freeze mask; frozen parameters must not move; the learning rate the fused AdamW sees follows the schedule."""
    from prismer_amd import schedules as S
    from prismer_amd.trainer import Trainer
    case = C.Case('tiny_caption')
    d = case.dims
    enc, dec, esd, dsd = build(case, p_drop=0.0)
    names = ['expert_encoder.' + n for n, _ in enc.named_parameters()] + ['text_decoder.' + n for n, _ in dec.named_parameters()]
    fm = O.freeze_mask(names, 'freeze_lang_vision')
    for n, p in enc.named_parameters():
        p.requires_grad = fm['expert_encoder.' + n]
    for n, p in dec.named_parameters():
        p.requires_grad = fm['text_decoder.' + n]
    assert not fm['text_decoder.roberta.encoder.layer.0.0.attention.self.query.weight'] and fm['text_decoder.roberta.encoder.layer.0.1.self.query.weight']
    m = _Holder(); m.expert_encoder, m.text_decoder = enc, dec
    x, ids, mask, _, _ = case.inputs()
    labels = ids.masked_fill(ids == d.pad_token_id, -100)           # prefix '' : nothing but the pads is masked
    tab = case.instance_table(x)
    sched = S.pretrain_schedule(steps_per_epoch=4, max_epoch=3, init_lr=3e-4, min_lr=1e-6, warmup_init_lr=1e-6, warmup_steps=3)
    tr = Trainer(m, lr=3e-4, weight_decay=0.05, use_graph=True, keep_grads=True, lr_schedule=sched)
    tr.set_batch(to_dev(x), ids, mask, labels)
    orig = tr._host_prologue

    def prologue():
        orig()
        tr.table.copy_(torch.tensor(tab, dtype=torch.int32))
    tr._host_prologue = prologue
    frozen0 = {n: p.detach().clone() for n, p in list(enc.named_parameters()) + list(dec.named_parameters()) if not p.requires_grad}
    loss = tr.step()
    torch.cuda.synchronize()
    assert abs(float(tr.hyper[0]) - sched(0)) < 1e-12 and sched(0) == 1e-6
    grads = {}
    for pref, st in (('expert_encoder.', tr.stores[0]), ('text_decoder.', tr.stores[1])):
        for nm in st.names:
            if st.is_trainable(nm):
                grads[pref + nm] = st.g(nm).detach().float().cpu().clone()
    # oracle
    leaves = {}
    for k, v in esd.items():
        if v.is_floating_point() and 'running' not in k and fm.get('expert_encoder.' + k, False):
            v.requires_grad_(True); leaves['expert_encoder.' + k] = v
    for k, v in dsd.items():
        if v.is_floating_point() and not k.startswith('lm_head.decoder.') and fm.get('text_decoder.' + k, False):
            v.requires_grad_(True); leaves['text_decoder.' + k] = v
    eo = O.vision_encoder(esd, x, d.patch_size, d.vit_heads, True, tab, {})
    _, ls = O.text_decoder(dsd, ids, mask, eo.transpose(0, 1), d.num_attention_heads, labels)
    ls.mean().backward()
    assert math_close(loss.item(), ls.mean().item(), TOL_LOSS)
    assert sorted(grads) == sorted(leaves)
    g = np.load(os.path.join(GOLD, 'tiny_caption.npz'))               # autocast yardstick of the same modules (freeze_vision fixture)
    bad = []
    for n, v in leaves.items():
        ref = v.grad
        if ref.norm() < 1e-4:
            continue
        e_norm = abs(grads[n].double().norm().item() - ref.double().norm().item()) / ref.double().norm().item()
        bar = max(TOL_GRAD, 2 * float(g['ac_norm.' + n])) if 'ac_norm.' + n in g else TOL_GRAD
        if e_norm > bar:
            bad.append((n, e_norm, bar))
    assert not bad, bad[:5]
    for _ in range(4):                                                  # steps 1..4: warm-up (1, 2), then its last value until the epoch ends
        tr.step()
    torch.cuda.synchronize()
    assert abs(float(tr.hyper[0]) - sched(4)) < 1e-9 and sched(3) == sched(2) and abs(sched(4) - S.cosine_lr(1, 3, 3e-4, 1e-6)) < 1e-15
    for n, p in list(enc.named_parameters()) + list(dec.named_parameters()):
        if n in frozen0:
            assert torch.equal(p.detach(), frozen0[n]), n


def test_prismer_huge_geometry_trains_one_step():
    """SURVEY 8f #4 / configs/prismer.json:50-73: the HUGE geometry (ViT-H/14: width 1280, 32 layers, 20 heads; Experts Resampler 8 heads x 160;
    roberta-large decoder cross-attending a 1280-wide stream) goes through the same kernels -- head dim 160 in the resampler, 1280-wide
    LayerNorm / stems, K = 1280 projections.  Shape smoke test: two native steps at batch 2, finite and decreasing loss, finite gradients."""
    import bench
    from prismer_amd.model.prismer_caption import PrismerCaption
    from prismer_amd.trainer import Trainer
    dims = config.prismer_huge()
    assert dims.width // dims.resampler_heads == 160 and dims.width // dims.vit_heads == 64 and dims.seq_len == 256 + 64
    torch.manual_seed(0)
    model = PrismerCaption({'experts': config.CAPTION_EXPERTS, 'image_resolution': 224, 'prismer_model': 'prismer_huge', 'freeze': 'freeze_vision'}).cuda()
    x, ids, mask, labels = bench.make_inputs(dims, 2, 16, 7, torch.device('cuda'))
    tr = Trainer(model, lr=2e-5, total_steps=10, use_graph=False, keep_grads=True)
    tr.set_batch(x, ids, mask, labels)
    l1 = tr.step().item()
    for st in tr.stores:
        g = st.grad[:st.n_train]
        assert torch.isfinite(g).all() and g.abs().max() > 0
    l2 = tr.step().item()
    assert np.isfinite(l1) and np.isfinite(l2) and l2 < l1, (l1, l2)
