"""Beam-search decoding for the caption / VQA heads (reference: text_decoder.generate(num_beams=3, ...) at
model/prismer_caption.py:45-50 and prismer_vqa.py:52-58, i.e. transformers' beam search with
prepare_inputs_for_generation of roberta.py:401-406: the FULL prefix is re-run every step, no KV cache).
Host-side control flow only; every decoder evaluation is the HIP forward program.  Implements the standard
length-normalised beam search (score = sum log p / len**length_penalty), min_length EOS suppression and
early stopping when `num_beams` finished hypotheses beat the best running beam."""
import torch


@torch.no_grad()
def beam_search(decoder, input_ids, attention_mask, enc, num_beams=3, max_length=20, min_length=8, eos_token_id=2, pad_token_id=1,
                length_penalty=1.0):
    B, T0 = input_ids.shape
    dev = input_ids.device
    nb = num_beams
    ids = input_ids.repeat_interleave(nb, dim=0)                       # [B*nb, T]
    att = attention_mask.repeat_interleave(nb, dim=0)
    enc_b = enc.repeat_interleave(nb, dim=0).contiguous()
    beam_scores = torch.zeros(B, nb, device=dev)
    beam_scores[:, 1:] = -1e9                                          # all beams start identical: keep one alive
    done = [[] for _ in range(B)]                                      # finished hypotheses (score, tokens)
    finished = [False] * B
    cur = T0
    while cur < max_length:
        out = decoder(ids, attention_mask=att, encoder_hidden_states=enc_b, return_dict=True)
        logp = torch.log_softmax(out.logits[:, -1, :].float(), dim=-1)                   # [B*nb, V]
        if cur < min_length:
            logp[:, eos_token_id] = -float('inf')
        V = logp.shape[-1]
        cand = (beam_scores.view(-1, 1) + logp).view(B, nb * V)
        top_s, top_i = cand.topk(2 * nb, dim=1)
        top_s, top_i = top_s.tolist(), top_i.tolist()
        new_ids, new_scores = [], []
        for b in range(B):
            nxt = []
            for s, i in zip(top_s[b], top_i[b]):
                beam, tok = divmod(i, V)
                row = b * nb + beam
                if tok == eos_token_id:
                    if len(nxt) < nb:                                  # only EOS candidates ranked inside the top nb count
                        done[b].append((s / ((cur + 1 - T0 + T0) ** length_penalty), ids[row].tolist() + [tok]))
                    continue
                nxt.append((s, row, tok))
                if len(nxt) == nb:
                    break
            while len(nxt) < nb:
                nxt.append((-1e9, b * nb, pad_token_id))
            if len(done[b]) >= nb:
                best_running = nxt[0][0] / ((cur + 1) ** length_penalty)
                worst_done = sorted(done[b], key=lambda t: -t[0])[nb - 1][0]
                if worst_done >= best_running:
                    finished[b] = True
            new_scores.append([n[0] for n in nxt])
            new_ids.append([(n[1], n[2]) for n in nxt])
        rows = torch.tensor([r for b in new_ids for r, _ in b], device=dev)
        toks = torch.tensor([t for b in new_ids for _, t in b], device=dev)
        ids = torch.cat([ids.index_select(0, rows), toks[:, None]], dim=1)
        att = torch.cat([att.index_select(0, rows), att.new_ones(B * nb, 1)], dim=1)
        beam_scores = torch.tensor(new_scores, device=dev)
        cur += 1
        if all(finished):
            break
    results = []
    for b in range(B):
        for k in range(nb):                                            # add the running beams as finished at max_length
            done[b].append((beam_scores[b, k].item() / (cur ** length_penalty), ids[b * nb + k].tolist()))
        results.append(torch.tensor(max(done[b], key=lambda t: t[0])[1], device=dev))
    return results
