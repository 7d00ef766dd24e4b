"""GPU: KV-cached decoding (reference model/prismer_caption.py:36-57, prismer_vqa.py:44-62, roberta.py:401-406).
  * cached step logits == full-prefix recompute logits (what the reference does every step);
  * beam search with the cache == beam search recomputing the prefix, token for token;
  * against the oracle: CPU fp32 decoder (oracle/prismer_oracle.py) under the loop-form transformers-4.26.1 beam search
    (oracle/beam_oracle.py) on the same weights / encoder states."""
import numpy as np
import pytest
import torch

from oracle import prismer_oracle as O
from oracle.beam_oracle import beam_search_loops
from tests.golden import cases as C
from tests.test_parity_gpu import build, to_dev
from tests.util import rel_fro

pytestmark = pytest.mark.gpu


def _setup(name, B=None):
    case = C.Case(name)
    enc, dec, esd, dsd = build(case)
    enc.eval(); dec.eval()
    x = case.inputs()[0]
    if B is not None:                                           # the first B items of the fixture's batch
        x = {k: ({kk: vv[:B] for kk, vv in v.items()} if isinstance(v, dict) else v[:B]) for k, v in x.items()}
    tab = case.instance_table(x)
    enc.instance_table = None if tab is None else torch.tensor(tab, dtype=torch.int32).cuda()
    with torch.no_grad():
        e = enc(to_dev(x)).permute(1, 0, 2).contiguous()
    return case, dec, esd, dsd, x, tab, e


def test_gather_rows_kernel():
    from prismer_amd import ops
    src = torch.randn(40, 6, 64, device='cuda').bfloat16()
    idx = torch.randint(0, 40, (40,), device='cuda', dtype=torch.int32)
    dst = torch.zeros_like(src)
    ops.gather_rows(src.view(40, 384), idx, dst.view(40, 384), cols=256)
    want = src[idx.long()].view(40, 384)
    assert torch.equal(dst.view(40, 384)[:, :256], want[:, :256]) and float(dst.view(40, 384)[:, 256:].abs().sum()) == 0.0


@pytest.mark.parametrize('name', ['tiny_caption', 'base_caption'])
def test_cached_decode_matches_full_prefix_recompute(name):
    case, dec, _, _, _, _, e = _setup(name)
    B = e.shape[0]
    d = case.dims
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, d.vocab_size, (B, 9), generator=g).cuda()
    ids[:, 0] = 0
    att = torch.ones_like(ids)
    if B > 1:
        att[1, 2] = 0; ids[1, 2] = d.pad_token_id             # a pad inside the prompt (VQA questions are padded to the longest)
    prog, enc_bf = dec.decoding_program(e)
    st = prog.decode_begin(enc_bf, 16)
    with torch.no_grad():
        T0 = 5
        lg = prog.decode(st, ids[:, :T0], att[:, :T0], 0).float()
        full = dec(ids[:, :T0], attention_mask=att[:, :T0], encoder_hidden_states=e, return_dict=True).logits[:, -1].float()
        assert rel_fro(lg, full) < 2e-3
        for T in range(T0 + 1, 10):
            lg = prog.decode(st, ids[:, :T], att[:, :T], T - 1).float()
            full = dec(ids[:, :T], attention_mask=att[:, :T], encoder_hidden_states=e, return_dict=True).logits[:, -1].float()
            assert rel_fro(lg, full) < 4e-3, T
            assert (lg.argmax(-1) == full.argmax(-1)).all()
        # beam reordering: caches follow the permutation
        rows = torch.arange(B - 1, -1, -1, device='cuda')
        prog.decode_reorder(st, rows)
        ids2, att2 = ids.index_select(0, rows), att.index_select(0, rows)
        enc_r = e.index_select(0, rows)
        st['kvs'] = [kv.view(B, -1, kv.shape[-1]).index_select(0, rows).reshape(-1, kv.shape[-1]) for kv in st['kvs']]
        nxt = torch.randint(3, d.vocab_size, (B, 1), generator=g).cuda()
        ids3 = torch.cat([ids2[:, :9], nxt], 1); att3 = torch.cat([att2[:, :9], att2.new_ones(B, 1)], 1)
        lg = prog.decode(st, ids3, att3, 9).float()
        full = dec(ids3, attention_mask=att3, encoder_hidden_states=enc_r, return_dict=True).logits[:, -1].float()
        assert rel_fro(lg, full) < 4e-3


@pytest.mark.parametrize('name,lp', [('tiny_caption', 1.0), ('tiny_vqa', -1.0), ('base_b8', 1.0)])
def test_beam_search_cached_equals_recompute_and_oracle(name, lp):
    """(round 3: also at the Prismer-BASE geometry, 4 images x 3 beams, full depth)"""
    from prismer_amd.model.generate import beam_search
    case, dec, _, dsd, _, _, e = _setup(name, B=4 if name == 'base_b8' else None)
    d = case.dims
    B = e.shape[0]
    prompt = torch.tensor([[0, 83 % d.vocab_size, 170 % d.vocab_size, 9]] * B).cuda()
    att = torch.ones_like(prompt)
    kw = dict(num_beams=3, max_length=14, min_length=8, eos_token_id=d.eos_token_id, pad_token_id=d.pad_token_id, length_penalty=lp)
    with torch.no_grad():
        fast = beam_search(dec, prompt, att, e, use_cache=True, **kw)
        slow = beam_search(dec, prompt, att, e, use_cache=False, **kw)
    if name == 'base_b8':
        # random-init BASE weights give near-uniform next-token distributions over 50 265 words: the bf16 noise between the cached
        # and the full-prefix decode can swap a near-tied continuation, so identity is required for all but one item here (the
        # token-for-token identity of the two decoders is pinned at the tiny geometry, where the distributions are peaked)
        assert sum(f.tolist() == s_.tolist() for f, s_ in zip(fast, slow)) >= B - 1, ([f.tolist() for f in fast], [s_.tolist() for s_ in slow])
    else:
        assert [f.tolist() for f in fast] == [s.tolist() for s in slow]
    out = dec.generate(input_ids=prompt, encoder_hidden_states=e, attention_mask=att, num_beams=3, max_length=14, min_length=8,
                       length_penalty=lp)
    assert out.shape[0] == B and [o[:len(f)].tolist() for o, f in zip(out, fast)] == [f.tolist() for f in fast]
    # oracle: fp32 CPU decoder under the loop-form beam search, full-prefix recompute like the reference
    enc_cpu = e.float().cpu()
    nb = 3

    def step_list(rows):
        ids = torch.tensor(rows)
        with torch.no_grad():
            lg, _ = O.text_decoder(dsd, ids, torch.ones_like(ids), enc_cpu.repeat_interleave(nb, 0), d.num_attention_heads)
        return torch.log_softmax(lg[:, -1].double(), -1).tolist()
    want = beam_search_loops(step_list, prompt.cpu().tolist(), nb, 14, 8, d.eos_token_id, d.pad_token_id, lp)
    same = sum(f.tolist() == w for f, w in zip(fast, want))
    print(name, 'hypotheses identical to the fp32 oracle:', same, 'of', B, [f.tolist() for f in fast], want)
    # bf16 logits vs fp32 logits can swap near-tied continuations; the chosen hypotheses must then score (under the ORACLE's
    # model) within bf16 noise of the oracle's own choice
    for f, w in zip(fast, want):
        if f.tolist() == w:
            continue
        def score(tokens):
            ids = torch.tensor([tokens])
            with torch.no_grad():
                lg, _ = O.text_decoder(dsd, ids, torch.ones_like(ids), enc_cpu[:1] if False else enc_cpu[[want.index(w)]], d.num_attention_heads)
            lp_ = torch.log_softmax(lg[0].double(), -1)
            s = sum(float(lp_[t - 1, tokens[t]]) for t in range(prompt.shape[1], len(tokens)))
            n = len(tokens) - (1 if tokens[-1] == d.eos_token_id else 0)
            return s / (n ** lp)
        assert abs(score(f.tolist()) - score(w)) < 0.05 * abs(score(w)) + 0.05, (f.tolist(), w)
    assert same >= 1
