"""Loop-form restatement of transformers 4.26.1 beam search (`generate(num_beams=k)` = BeamSearchScorer.process / finalize +
BeamHypotheses + MinLengthLogitsProcessor)  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference calls it at model/prismer_caption.py:45-50 (num_beams 3, max_length 20, min_length 8) and
model/prismer_vqa.py:52-58 (length_penalty -1).  transformers is a third-party dependency pinned at ~=4.26.1
(requirements.txt:6) and its beam search cannot run on the reference decoder under the installed 5.x (no GenerationMixin on
that class), so this file restates the published 4.26.1 algorithm:

  generation/beam_search.py  BeamSearchScorer.process   -- visit the 2*num_beams best (beam, token) continuations of an item
                             in score order; EOS continuations ranked < num_beams become hypotheses scored
                             sum_logprobs / len(prompt+generated so far) ** length_penalty; others refill the beams until
                             num_beams are collected; then `done |= hyps.is_done(best_candidate_score, cur_len)`
                             BeamHypotheses.add / is_done  -- keep the num_beams best; done when the worst kept one is at
                             least best_candidate_score / cur_len ** length_penalty (early_stopping=False)
                             BeamSearchScorer.finalize    -- running beams of unfinished items become hypotheses; best one wins
  generation/logits_process.py  MinLengthLogitsProcessor -- EOS score = -inf while cur_len < min_length

Parity status: UNPINNED against transformers itself (4.26.1 is not installable here; 5.x changed the length normalisation to
exclude the prompt).  It pins the vectorised device implementation (prismer_amd/model/generate.py) in tests/test_generate_cpu.py.
"""
import math


def beam_search_loops(step_fn, prompt, num_beams, max_length, min_length, eos, pad, length_penalty=1.0):
    """step_fn(list of token lists) -> list of per-row log-probabilities (list of floats, len V).  prompt: list of B token
    lists of equal length.  Returns a list of B token lists."""
    B, nb = len(prompt), num_beams
    rows = [list(p) for p in prompt for _ in range(nb)]
    scores = [[0.0] + [-1e9] * (nb - 1) for _ in range(B)]
    hyps = [[] for _ in range(B)]                  # (score, tokens)
    done = [False] * B
    cur = len(prompt[0])
    while cur < max_length:
        logp = step_fn(rows)
        V = len(logp[0])
        new_rows, new_scores = [], []
        for b in range(B):
            if done[b]:
                new_rows += [rows[b * nb] + [pad]] * nb
                new_scores.append([0.0] * nb)
                continue
            cand = []
            for k in range(nb):
                lp = list(logp[b * nb + k])
                if cur < min_length:
                    lp[eos] = -math.inf
                for t in range(V):
                    cand.append((scores[b][k] + lp[t], k, t))
            cand.sort(key=lambda c: -c[0])
            cand = cand[:2 * nb]
            nxt = []
            for rank, (s, k, t) in enumerate(cand):
                if t == eos:
                    if rank >= nb:
                        continue
                    h = (s / (cur ** length_penalty), rows[b * nb + k] + [eos])
                    if len(hyps[b]) < nb or h[0] > min(x[0] for x in hyps[b]):
                        hyps[b].append(h)
                        if len(hyps[b]) > nb:
                            hyps[b].remove(min(hyps[b], key=lambda x: x[0]))
                else:
                    nxt.append((s, k, t))
                if len(nxt) == nb:
                    break
            assert len(nxt) == nb
            if len(hyps[b]) >= nb and min(x[0] for x in hyps[b]) >= cand[0][0] / (cur ** length_penalty):
                done[b] = True
            new_rows += [rows[b * nb + k] + [t] for _, k, t in nxt]
            new_scores.append([s for s, _, _ in nxt])
        rows, scores = new_rows, new_scores
        cur += 1
    out = []
    for b in range(B):
        if not done[b]:
            for k in range(nb):
                h = (scores[b][k] / (cur ** length_penalty), rows[b * nb + k])
                if len(hyps[b]) < nb or h[0] > min(x[0] for x in hyps[b]):
                    hyps[b].append(h)
                    if len(hyps[b]) > nb:
                        hyps[b].remove(min(hyps[b], key=lambda x: x[0]))
        out.append(max(hyps[b], key=lambda x: x[0])[1])
    return out
