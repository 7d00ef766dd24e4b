"""Known-answer test of the dropout generator.  The kernels draw dropout masks from Philox4x32 (csrc/common.h: philox4x32, PH_PHILOX_ROUNDS
rounds); tests/util.py mirrors it in numpy and the GPU tests compare kernel masks with the mirror.  This test pins the MIRROR to the published
known-answer vectors of the Random123 library (kat_vectors, philox4x32 with 7 and 10 rounds: counter and key all zero, all ones, digits of pi),
so that "mask = a pure function of (seed, stream, element index)" rests on the published generator and not on two copies of one mistake, and it
pins the round count of the mirror to the kernel source."""
import os
import re

import numpy as np
import pytest

from tests import util

KAT = [   # rounds, counter (4 words), key (2 words), expected output (4 words)
    (10, (0, 0, 0, 0), (0, 0), '6627e8d5 e169c58d bc57ac4c 9b00dbd8'),
    (10, (0xffffffff,) * 4, (0xffffffff, 0xffffffff), '408f276d 41c83b0e a20bc7c6 6d5451fd'),
    (10, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), 'd16cfe09 94fdcceb 5001e420 24126ea1'),
    (7, (0, 0, 0, 0), (0, 0), '5f6fb709 0d893f64 4f121f81 4f730a48'),
]


@pytest.mark.parametrize('rounds,ctr,key,want', KAT)
def test_numpy_mirror_reproduces_random123_vectors(rounds, ctr, key, want, monkeypatch):
    monkeypatch.setattr(util, 'PHILOX_ROUNDS', rounds)
    c = [np.array([x], dtype=np.uint32) for x in ctr]
    out = util.philox4x32(c[0], c[1], c[2], c[3], key[0], key[1])
    assert ' '.join(f'{int(x[0]):08x}' for x in out) == want


def test_round_count_matches_the_kernel_source():
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'prismer_amd', 'csrc', 'common.h')).read()
    m = re.search(r'#define PH_PHILOX_ROUNDS_N (\d+)', src)
    assert m and int(m.group(1)) == util.PHILOX_ROUNDS == 7
    assert 'for (int r = 0; r < PH_PHILOX_ROUNDS; ++r) {' in src
    # the round function itself: multipliers, Weyl key increments and the word permutation
    for frag in ('__umulhi(0xD2511F53u, c0)', '__umulhi(0xCD9E8D57u, c2)', 'n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0',
                 'k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;'):
        assert frag in src


def test_keep_rate_of_the_seven_round_generator():
    """dropout_keep_linear over 2^20 elements: keep rate within 4 sigma of 1 - p, and no correlation between neighbouring words"""
    n, p = 1 << 20, 0.1
    keep = util.dropout_keep_linear(n, seed=0x1234567890abcdef, stream=77, p=p).numpy().astype(np.float64)
    rate = keep.mean()
    assert abs(rate - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n), rate
    a, b = keep[:-1] - rate, keep[1:] - rate
    assert abs((a * b).mean() / (p * (1 - p))) < 5e-3
