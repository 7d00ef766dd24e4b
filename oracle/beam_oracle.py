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

Parity status: beam_search_loops (4.26.1 semantics) cannot be run against 4.26.1 itself (not installable here, no source on this
machine).  What IS pinned is the method: beam_search_loops_v5 below is the same loop-form reading applied to the source of the
installed release (5.15.0) and reproduces that release's `generate(num_beams=k)` token for token on 45 random causal LMs with
EOS-terminated hypotheses and early stops (tests/test_oracle_beam_vs_transformers.py); beam_search_loops differs from it in the
three places listed in its docstring (length normalisation, max_length handling, early-stop rule -- the documented changes between
the releases) and agrees with it exactly where those cannot matter (same test file).  beam_search_loops in turn pins the vectorised
device implementation (prismer_amd/model/generate.py) in tests/test_generate_cpu.py / test_generate_gpu.py.
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


def beam_search_loops_v5(step_fn, prompt, num_beams, max_length, min_length, eos, pad, length_penalty=1.0):
    """Loop-form restatement of the beam search of the transformers release that IS installed here (5.15.0:
    generation/utils.py GenerationMixin._beam_search with _get_top_k_continuations, _get_running_beams_for_next_iteration,
    _update_finished_beams, _check_early_stop_heuristic, _beam_search_has_unfinished_sequences; early_stopping=False, one EOS id).
    It exists to PIN THE METHOD: tests/test_oracle_beam_vs_transformers.py drives `generate(num_beams=k)` of that release on small
    random causal LMs and requires this function to reproduce it token for token.  Differences to beam_search_loops (4.26.1,
    the release the reference pins, requirements.txt:6) -- the only places where the two functions differ:
      1. hypothesis score: sum_logprobs / (cur_len + 1 - prompt_len) ** lp here (generated tokens incl. EOS),
         sum_logprobs / cur_len ** lp there (prompt + generated tokens before the EOS);
      2. reaching max_length is a stopping criterion here: in the step that fills position max_length - 1 every top-num_beams
         continuation becomes a hypothesis; there the loop ends and BeamSearchScorer.finalize adds the running beams of the items
         that are not done, scored with the full length;
      3. the early-stop test compares the worst kept hypothesis with best RUNNING beam / (cur_len - prompt_len) ** lp after the step
         here, with the best of all 2k candidates / cur_len ** lp before the step there; here the loop runs until no item can
         improve (or max_length), a finished item just stops accepting hypotheses.
    Same in both: 2k candidates per item in score order, only candidates ranked < num_beams may become hypotheses, num_beams best
    hypotheses kept, MinLengthLogitsProcessor."""
    B, nb = len(prompt), num_beams
    plen = len(prompt[0])
    rows = [list(p) for p in prompt for _ in range(nb)]
    run_scores = [[0.0] + [-1e9] * (nb - 1) for _ in range(B)]
    fin = [[(-1e9, None, False)] * nb for _ in range(B)]           # (score, tokens, is_finished), sorted best first
    unsat = [True] * B
    cur = plen
    while True:
        logp = step_fn(rows)
        V = len(logp[0])
        all_hit = True
        new_rows, new_scores = [], []
        for b in range(B):
            cand = []
            for k in range(nb):
                lp = list(logp[b * nb + k])
                if cur < min_length:
                    lp[eos] = -math.inf
                for t in range(V):
                    cand.append((run_scores[b][k] + lp[t], k, t))
            cand.sort(key=lambda c: -c[0])
            cand = cand[:2 * nb]
            hit = [t == eos or cur + 1 >= max_length for _, _, t in cand]
            all_hit = all_hit and all(hit)
            # e. running beams of the next step: best num_beams continuations that did not stop
            masked = sorted(((s + (-1e9 if h else 0.0), i) for i, ((s, _, _), h) in enumerate(zip(cand, hit))), key=lambda x: -x[0])[:nb]
            new_rows += [rows[b * nb + cand[i][1]] + [cand[i][2]] for _, i in masked]
            new_scores.append([s for s, _ in masked])
            # f. finished beams: stopped continuations ranked < num_beams, while the item can still improve
            merged = list(fin[b])
            for i, ((s, k, t), h) in enumerate(zip(cand, hit)):
                ok = h and i < nb and unsat[b]
                sc = s / ((cur + 1 - plen) ** length_penalty) if ok else s / ((cur + 1 - plen) ** length_penalty) - 1e9
                merged.append((sc, rows[b * nb + k] + [t], ok))
            merged.sort(key=lambda x: -x[0])
            fin[b] = merged[:nb]
        rows, run_scores = new_rows, new_scores
        cur += 1
        for b in range(B):
            full = all(f[2] for f in fin[b])
            worst = min(f[0] for f in fin[b]) if full else -1e9
            best = run_scores[b][0] / ((cur - plen) ** length_penalty)
            unsat[b] = unsat[b] and (not full or best > worst)
        if not any(unsat) or all_hit:
            break
    return [fin[b][0][1] for b in range(B)]
