"""Beam-search decoding for the caption / VQA heads.

Reference: `self.text_decoder.generate(input_ids=..., encoder_hidden_states=..., num_beams=3, max_length=..., min_length=...,
length_penalty=...)` at model/prismer_caption.py:45-50 and model/prismer_vqa.py:52-58, i.e. transformers' (4.26.1)
beam search: BeamSearchScorer.process / finalize, BeamHypotheses.add / is_done, MinLengthLogitsProcessor.  The reference re-runs
the FULL prefix every step (prepare_inputs_for_generation, roberta.py:401-406, passes no cache).

Here:
  * the decoder runs KV-cached (DecoderProgram.decode_begin / decode / decode_reorder): cross-attention K/V projected once,
    self-attention K/V cached per layer and re-ordered with the beams on the device;
  * the beam bookkeeping is vectorised over the batch with tensor ops on the device the logits live on -- no `.tolist()` /
    host round trip per step (the only synchronisation is the final read-out);
  * `beam_search_from_logits` is the algorithm alone (any callable that maps the current ids to next-token logits): it runs on
    CPU tensors too, which is how tests/test_generate_cpu.py checks it against the loop-form oracle (oracle/beam_oracle.py).

Semantics kept from transformers 4.26.1 (num_return_sequences = 1, early_stopping = False, do_sample = False):
  - scores of the first step come from beam 0 only (the other beams start at -1e9);
  - EOS is banned while cur_len < min_length; cur_len counts the decoder prompt;
  - per step the best 2*num_beams (beam, token) candidates are visited in score order until num_beams continuing beams are
    found; an EOS candidate becomes a finished hypothesis only if its rank is < num_beams, with score
    sum_logprobs / cur_len ** length_penalty (cur_len = length before the EOS); a batch item keeps its num_beams best hypotheses;
  - an item is done when it holds num_beams hypotheses and the worst of them is at least best_running / cur_len ** length_penalty;
    done items are frozen (padding beams);
  - at max_length the running beams of unfinished items are added as hypotheses; the best hypothesis wins.
"""
import torch

NEG = -1e9


@torch.no_grad()
def beam_search_from_logits(step_fn, input_ids, num_beams=3, max_length=20, min_length=8, eos_token_id=2, pad_token_id=1,
                            length_penalty=1.0, reorder_fn=None):
    """step_fn(ids [B*nb, T]) -> next-token logits [B*nb, V] (any float dtype) for the LAST position of every row.
    reorder_fn(rows int64 [B*nb]) is called after each step with the source row of every continuing beam (cache re-ordering).
    Returns a list of B 1-D LongTensors (prompt + generated tokens, EOS included when the hypothesis ended with one)."""
    B, T0 = input_ids.shape
    dev = input_ids.device
    nb = num_beams
    Tmax = max_length
    ids = torch.full((B * nb, Tmax), pad_token_id, dtype=torch.long, device=dev)
    ids[:, :T0] = input_ids.repeat_interleave(nb, dim=0)
    beam_scores = torch.zeros(B, nb, device=dev)
    beam_scores[:, 1:] = NEG
    hyp_score = torch.full((B, nb), -float('inf'), device=dev)            # finished hypotheses (BeamHypotheses): score,
    hyp_tok = torch.full((B, nb, Tmax), pad_token_id, dtype=torch.long, device=dev)      # tokens,
    hyp_len = torch.zeros(B, nb, dtype=torch.long, device=dev)            # length (EOS included)
    n_hyp = torch.zeros(B, dtype=torch.long, device=dev)
    done = torch.zeros(B, dtype=torch.bool, device=dev)
    ar_b = torch.arange(B, device=dev)
    cur = T0

    def add_hyp(sel, score, tokens, length):
        """BeamHypotheses.add for the batch items in `sel` (bool [B]): append while fewer than nb, else replace the worst if better"""
        worst, worst_i = hyp_score.min(dim=1)
        slot = torch.where(n_hyp < nb, n_hyp.clamp(max=nb - 1), worst_i)
        ok = sel & ((n_hyp < nb) | (score > worst))
        b = ar_b[ok]
        s = slot[ok]
        hyp_score[b, s] = score[ok]
        hyp_tok[b, s] = tokens[ok]
        hyp_len[b, s] = length[ok]
        n_hyp[ok & (n_hyp < nb)] += 1

    T_start = cur
    while cur < Tmax:
        logits = step_fn(ids[:, :cur])
        logp = torch.log_softmax(logits.float(), dim=-1)
        V = logp.shape[-1]
        if cur < min_length:
            logp[:, eos_token_id] = -float('inf')
        cand = (beam_scores.view(-1, 1) + logp).view(B, nb * V)
        top_s, top_i = cand.topk(2 * nb, dim=1)                          # sorted, best first
        top_beam, top_tok = top_i // V, top_i % V
        n_next = torch.zeros(B, dtype=torch.long, device=dev)
        nxt_score = torch.zeros(B, nb, device=dev)
        nxt_row = (ar_b * nb)[:, None].repeat(1, nb)                     # padding beams of done items: (score 0, pad, first row)
        nxt_tok = torch.full((B, nb), pad_token_id, dtype=torch.long, device=dev)
        cur_ids = ids.view(B, nb, Tmax)
        for r in range(2 * nb):
            s, bm, tk = top_s[:, r], top_beam[:, r], top_tok[:, r]
            live = ~done & (n_next < nb)                                 # the reference loop breaks once nb beams are collected
            is_eos = tk == eos_token_id
            if r < nb:                                                   # EOS ranked inside the top nb: a finished hypothesis
                e = live & is_eos
                tokens = cur_ids[ar_b, bm].clone()
                tokens[:, cur] = eos_token_id
                add_hyp(e, s / (cur ** length_penalty), tokens, torch.full((B,), cur + 1, dtype=torch.long, device=dev))
            c = live & ~is_eos
            b = ar_b[c]
            k = n_next[c]
            nxt_score[b, k] = s[c]
            nxt_row[b, k] = (b * nb + bm[c])
            nxt_tok[b, k] = tk[c]
            n_next[c] += 1
        # is_done (4.26.1): num_beams hypotheses held and the worst one beats what the best running beam could still score
        best_running = top_s[:, 0] / (cur ** length_penalty)
        done = done | ((n_hyp >= nb) & (hyp_score.min(dim=1).values >= best_running))
        rows = nxt_row.reshape(-1)
        ids = ids.index_select(0, rows)
        ids[:, cur] = nxt_tok.reshape(-1)
        beam_scores = nxt_score
        if reorder_fn is not None:
            reorder_fn(rows)
        cur += 1
        # transformers breaks out once every item is done (beam_scorer.is_done); one small device->host sync every 4 steps here
        if (cur - T_start) % 4 == 0 and bool(done.all()):
            break
    # finalize: running beams of unfinished items become hypotheses (score / cur_len ** lp)
    cur_ids = ids.view(B, nb, Tmax)
    for k in range(nb):
        add_hyp(~done, beam_scores[:, k] / (cur ** length_penalty), cur_ids[:, k], torch.full((B,), cur, dtype=torch.long, device=dev))
    best = hyp_score.argmax(dim=1)
    out_tok = hyp_tok[ar_b, best].cpu()
    out_len = hyp_len[ar_b, best].cpu()
    return [out_tok[b, :int(out_len[b])].to(dev) for b in range(B)]


@torch.no_grad()
def beam_search(decoder, input_ids, attention_mask, enc, num_beams=3, max_length=20, min_length=8, eos_token_id=2, pad_token_id=1,
                length_penalty=1.0, use_cache=True):
    """decoder: prismer_amd RobertaForCausalLMModified (eval mode); enc: [B, S, Hv] encoder states (one row per image).
    use_cache=False re-runs the full prefix every step like the reference (kept for the equivalence test)."""
    B, T0 = input_ids.shape
    nb = num_beams
    enc_b = enc.repeat_interleave(nb, dim=0).contiguous()
    att0 = attention_mask.repeat_interleave(nb, dim=0)

    def mask_for(T):                                               # generated positions are real tokens
        if T == T0:
            return att0
        return torch.cat([att0, att0.new_ones(B * nb, T - T0)], dim=1)
    if not use_cache:
        def step(ids):
            out = decoder(ids, attention_mask=mask_for(ids.shape[1]), encoder_hidden_states=enc_b, return_dict=True)
            return out.logits[:, -1, :]
        return beam_search_from_logits(step, input_ids, nb, max_length, min_length, eos_token_id, pad_token_id, length_penalty)
    prog, enc_bf = decoder.decoding_program(enc_b)
    st = prog.decode_begin(enc_bf, max_length)

    def step(ids):
        T = ids.shape[1]
        return prog.decode(st, ids, mask_for(T), 0 if T == T0 else T - 1)
    return beam_search_from_logits(step, input_ids, nb, max_length, min_length, eos_token_id, pad_token_id, length_penalty,
                                   reorder_fn=lambda rows: prog.decode_reorder(st, rows))
