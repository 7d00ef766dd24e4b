"""CPU: the vectorised beam search of prismer_amd/model/generate.py (tensor ops only -- it runs wherever its logits live)
against the loop-form oracle of transformers-4.26.1 beam search (oracle/beam_oracle.py), on synthetic next-token tables that
provoke early EOS, min_length suppression, finished items and both length-penalty signs."""
import math

import pytest
import torch

from oracle.beam_oracle import beam_search_loops
from prismer_amd.model.generate import beam_search_from_logits


def make_lm(V, seed, eos, eos_boost):
    """a deterministic toy LM: logits depend on the last two tokens and the position"""
    g = torch.Generator().manual_seed(seed)
    W1 = torch.randn(V, V, generator=g)
    W2 = torch.randn(V, V, generator=g) * 0.5
    Wp = torch.randn(64, V, generator=g) * 0.3

    def logits(ids):                                   # ids: LongTensor [R, T]
        T = ids.shape[1]
        z = W1[ids[:, -1]] + (W2[ids[:, -2]] if T > 1 else 0) + Wp[T % 64]
        z = z.clone()
        z[:, eos] += eos_boost
        return z
    return logits


@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('lp,eos_boost', [(1.0, 0.0), (1.0, 2.5), (-1.0, 1.0), (0.5, 4.0)])
def test_vectorised_beam_search_matches_loop_oracle(seed, lp, eos_boost):
    V, B, nb, T0, Tmax, Tmin, eos, pad = 23, 4, 3, 3, 12, 6, 2, 1
    lm = make_lm(V, seed, eos, eos_boost)
    g = torch.Generator().manual_seed(1000 + seed)
    prompt = torch.randint(3, V, (B, T0), generator=g)
    prompt[:, 0] = 0
    calls = []

    def step(ids):
        calls.append(ids.shape)
        return lm(ids)
    got = beam_search_from_logits(step, prompt, nb, Tmax, Tmin, eos, pad, lp)

    def step_list(rows):
        z = lm(torch.tensor(rows))
        return torch.log_softmax(z.float(), -1).tolist()
    want = beam_search_loops(step_list, prompt.tolist(), nb, Tmax, Tmin, eos, pad, lp)
    assert [g_.tolist() for g_ in got] == want
    assert all(len(w) <= Tmax for w in want) and len(calls) <= Tmax - T0      # (round 3: the loop leaves once every item is done, like transformers)
    assert all((eos not in w[:Tmin]) for w in want)                        # min_length: no EOS before position Tmin


def test_reorder_callback_sees_every_beam_permutation():
    V, B, nb, eos, pad = 11, 2, 3, 2, 1
    lm = make_lm(V, 3, eos, 0.0)
    prompt = torch.tensor([[0, 5, 6], [0, 7, 8]])
    rows_seen = []
    beam_search_from_logits(lm, prompt, nb, 8, 4, eos, pad, 1.0, reorder_fn=lambda r: rows_seen.append(r.clone()))
    assert len(rows_seen) == 8 - 3 and all(r.shape == (B * nb,) for r in rows_seen)
    for r in rows_seen:                                                    # a beam only ever continues a beam of ITS item
        assert ((r // nb) == torch.arange(B).repeat_interleave(nb)).all()
