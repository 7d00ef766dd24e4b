"""The loop-form beam-search restatements of oracle/beam_oracle.py against a REAL generator.

The reference calls transformers-4.26.1 `generate(num_beams=3)` (model/prismer_caption.py:45-50, prismer_vqa.py:52-58); that release
is not installable here and the installed one (5.x) re-implemented beam search.  What can be pinned is the METHOD: the same
loop-form reading applied to the installed release's source (beam_search_loops_v5) must reproduce that release's `generate` token
for token on random causal LMs.  beam_search_loops (4.26.1) differs from it in the three places its docstring lists, and nowhere
else -- the last test checks that on inputs where those three places cannot matter."""
import math

import pytest
import torch

from oracle.beam_oracle import beam_search_loops, beam_search_loops_v5

transformers = pytest.importorskip('transformers')
COVER = {'eos': 0, 'early': 0}


def _tiny_lm(seed, vocab, eos_bias=0.0):
    from transformers import GPT2Config, GPT2LMHeadModel
    torch.manual_seed(seed)
    cfg = GPT2Config(vocab_size=vocab, n_positions=32, n_embd=16, n_layer=1, n_head=2, bos_token_id=0, eos_token_id=1, pad_token_id=2)
    m = GPT2LMHeadModel(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(4.0)                                        # peaked next-token distributions
    if eos_bias:                                               # EOS competitive at every step: hypotheses early, the early-stop rule fires

        def hook(mod, inp, out):
            out = out.clone()
            out[..., 1] += eos_bias
            return out
        m.lm_head.register_forward_hook(hook)
    return m


CASES = [(s, v, nb, lp, eb) for s in range(5) for v, nb, lp in ((12, 3, 1.0), (9, 2, 1.0), (12, 3, -1.0)) for eb in (0.0, 3.0, 6.0)]


@pytest.mark.parametrize('seed,vocab,nb,lp,eos_bias', CASES)
def test_v5_loops_reproduce_installed_generate(seed, vocab, nb, lp, eos_bias):
    m = _tiny_lm(seed, vocab, eos_bias)
    g = torch.Generator().manual_seed(100 + seed)
    B, plen, Tmax, Tmin = 3, 3, 11, 5
    prompt = torch.randint(3, vocab, (B, plen), generator=g)
    with torch.no_grad():
        out = m.generate(input_ids=prompt, attention_mask=torch.ones_like(prompt), num_beams=nb, max_length=Tmax, min_length=Tmin,
                         length_penalty=lp, early_stopping=False, do_sample=False, eos_token_id=1, pad_token_id=2, use_cache=True)

    calls = []

    def step(rows):
        calls.append(len(rows))
        with torch.no_grad():
            logits = m(input_ids=torch.tensor(rows)).logits[:, -1].float()
        return torch.log_softmax(logits, -1).tolist()
    want = beam_search_loops_v5(step, prompt.tolist(), nb, Tmax, Tmin, 1, 2, lp)
    COVER['eos'] += sum(w[-1] == 1 and len(w) < Tmax for w in want)
    COVER['early'] += len(calls) < Tmax - plen
    for b in range(B):
        got = out[b].tolist()
        while len(got) > len(want[b]) and got[-1] == 2:
            got.pop()
        assert got == want[b], (b, got, want[b])


def test_the_cases_above_exercise_eos_and_early_stopping():
    """(runs after the parametrised cases) hypotheses that end in EOS before max_length, and searches that stop before max_length"""
    assert COVER['eos'] >= 20 and COVER['early'] >= 5, COVER


def test_both_restatements_agree_where_their_differences_cannot_matter():
    """No EOS before max_length (its log-probability is -inf until the last step) and length_penalty 0: no hypothesis exists before
    the last step, so neither early-stop rule can fire, normalisation is the identity, and `finalize` (4.26.1) / the max_length
    stopping criterion (5.x) both pick the best running beam -- candidate ranking and beam bookkeeping are all that is left."""
    g = torch.Generator().manual_seed(7)
    V, nb, B, plen, Tmax = 10, 3, 4, 2, 9
    table = torch.log_softmax(torch.randn(64, V, generator=g) * 2.0, -1)

    def step(rows):
        out = []
        for r in rows:
            lp = table[(sum((i + 1) * t for i, t in enumerate(r)) * 31 + len(r)) % 64].clone()
            lp[1] = -math.inf
            out.append(lp.tolist())
        return out
    prompt = torch.randint(3, V, (B, plen), generator=g).tolist()
    a = beam_search_loops(step, prompt, nb, Tmax, 0, 1, 2, 0.0)
    b = beam_search_loops_v5(step, prompt, nb, Tmax, 0, 1, 2, 0.0)
    assert a == b
