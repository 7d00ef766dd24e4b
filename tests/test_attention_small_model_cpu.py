"""Model check of the index algebra of the small-query attention backward (prismer_amd/csrc/attention.hip, attn_bwd_small_kernel, round 5).

The kernel computes P~ and dS in the dQ orientation (MFMA 16x16x32 C layout of S^T = K Q^T: lane (c, g) holds query column t*16 + c and the keys
nt*16 + g*4 + r of a 32-key unit), writes them through a per-wave LDS scratch [key][query] and reads them back as the B operands of
dV^T += dO^T P~ and dK^T += Q^T dS, whose A operands come from `frag_tr` (transposing LDS reads of the dO / Q images).  An MFMA contracts element j
of A with element j of B for j = 0..7 and the four lane groups g = 0..3, so both operands must enumerate the SAME query for every (g, j).  A wrong
index computes plausible garbage; the GPU tests catch it numerically, this test pins the algebra itself:
  * writer -> scratch -> reader delivers, to lane (c', g') element j, the value of (key kt*16 + c', query kappa(g', j));
  * `frag_tr` delivers, to lane (i, g') element j, row kappa(g', j) of the image -- the same kappa;
  * every (key, query) pair of a unit is written exactly once and read exactly once per 16-key sub-tile;
  * the scratch row stride (SM_TS = 36 bf16) keeps the 2-byte writes of a wave instruction on distinct 4-byte bank words per (g, r) row group."""
import os
import re

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'prismer_amd', 'csrc', 'attention.hip')
SM_TS, QT = 36, 2


def kappa(g, j):
    """key / query order of two adjacent 16-row C tiles as one 32-deep MFMA k-step (attention.hip, frag_tr comment)"""
    return 16 * (j >> 2) + 4 * g + (j & 3)


def writer_elements(lane):
    """(scratch index, (key, query)) pairs lane `lane` writes for one unit: for t, nt, r -> tsc[key * SM_TS + t * 16 + c]"""
    c, g = lane & 15, lane >> 4
    out = []
    for t in range(QT):
        for nt in range(2):
            for r in range(4):
                key = nt * 16 + g * 4 + r
                out.append((key * SM_TS + t * 16 + c, (key, t * 16 + c)))
    return out


def reader_elements(lane, kt):
    """what lane `lane` packs into its B operand for the 16-key sub-tile kt: row = tsc + (kt*16 + c) * SM_TS + 4*g; p0 = row[0:4], p1 = row[16:20]"""
    c, g = lane & 15, lane >> 4
    row = (kt * 16 + c) * SM_TS + 4 * g
    return [row + j for j in range(4)] + [row + 16 + j for j in range(4)]


def frag_tr_rows(lane, kbase=0):
    """image rows lane (i, g) receives from frag_tr(img, kbase, col0, lane), element j = 0..7"""
    g = lane >> 4
    return [kbase + kappa(g, j) for j in range(8)]


def test_scratch_round_trip_enumerates_queries_like_frag_tr():
    scratch = {}
    for lane in range(64):
        for idx, tag in writer_elements(lane):
            assert idx not in scratch, f'two lanes write scratch[{idx}]'
            scratch[idx] = tag
    assert len(scratch) == 32 * 32                                    # 32 keys x 32 queries, each written once
    seen = set()
    for kt in range(2):
        for lane in range(64):
            c, g = lane & 15, lane >> 4
            got = [scratch[i] for i in reader_elements(lane, kt)]
            want_rows = frag_tr_rows(lane)                             # the A operand's query order for this lane group
            for j, (key, query) in enumerate(got):
                assert key == kt * 16 + c, (lane, kt, j, key)
                assert query == kappa(g, j) == want_rows[j], (lane, kt, j, query)
                seen.add((key, query))
    assert len(seen) == 32 * 32                                        # every pair reaches exactly one (lane, element)


def test_scratch_writes_spread_over_banks():
    """one wave store instruction = fixed (t, nt, r), all 64 lanes: 16 consecutive bf16 per lane group g on key row nt*16 + g*4 + r.  With a 72-B row
    stride the four groups' rows start 8 bank words apart (4 rows x 18 words = 72 = 8 mod 64): the 4 x 8 words of the instruction are distinct"""
    for t in range(QT):
        for nt in range(2):
            for r in range(4):
                words = set()
                for lane in range(64):
                    c, g = lane & 15, lane >> 4
                    byte = ((nt * 16 + g * 4 + r) * SM_TS + t * 16 + c) * 2
                    words.add((byte // 4) % 64)
                assert len(words) == 32, (t, nt, r, len(words))        # 64 lanes x 2 B = 32 distinct 4-byte words, no two on one bank


def test_model_matches_the_sources():
    s = open(SRC).read()
    assert 'constexpr int SM_TS = 36;' in s
    assert 'tsc[key * SM_TS + t * 16 + c] = f2bf(qi[t] < f.Sq ? pdrop : 0.f);' in s
    assert 'tsc[32 * SM_TS + key * SM_TS + t * 16 + c] = f2bf(qi[t] < f.Sq ? dsv : 0.f);' in s
    assert 'const int key = nt * 16 + g * 4 + r;' in s
    assert 'const bf16* row = tsc + (kt * 16 + c) * SM_TS + 4 * g;' in s
    assert re.search(r'p0 = \*reinterpret_cast<const bf16x4\*>\(row\), p1 = \*reinterpret_cast<const bf16x4\*>\(row \+ 16\);', s)
    assert 'kappa(g,j) = kbase + 16*(j>>2) + 4*g + (j&3)' in s          # frag_tr's documented order
    assert 'const bf16x8 dfr = frag_tr<DH>(doimg, 0, d * 16, lane);' in s and 'const bf16x8 qfr = frag_tr<DH>(qimg, 0, d * 16, lane);' in s
