"""Model check of the block -> tile mapping of the 256x128 LDS-DMA GEMM kernel (prismer_amd/csrc/gemm_big.hip, big_tile, XCD_REMAP = true;
round 4: "XCD panel ownership").

Hardware places block b of a 1-D grid on XCD b % 8, and every XCD has its own L2.  The kernel therefore gives XCD x a run of whole 256-row
panels (tiles_m / 8 of them, one more for the first tiles_m % 8 XCDs) with ALL their column tiles, so that a row panel of the A operand is
fetched by one L2 only; the grid is 8 x (largest share) x tiles_n and the blocks beyond an XCD's share exit.  A wrong mapping computes some
tile twice and another never -- the GPU tests would catch that on the shapes they run; this test checks the index algebra for every small
(tiles_m, tiles_n) and the properties the traffic accounting of DESIGN.md rests on:
  * every tile is computed by exactly one block, idle blocks are exactly the surplus of the grid;
  * all column tiles of a row panel are computed on ONE XCD, and an XCD's panels are consecutive;
  * the grouped convolution launch numbers every problem like a single launch, at block offsets that are multiples of 8 (so that
    "block id mod 8" is still the XCD inside a problem)."""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'prismer_amd', 'csrc')
GM = 4


def panels(tiles_m, tiles_n):
    """gemm_common.h big_xcd_panels: panel ownership iff every XCD owns a panel and the largest share needs no more rounds of an XCD's 32 CUs
    than an even split of the tile list (round-4 advisor finding: tiles_m < 8 left XCDs idle, tiles_m = 8k + 1 doubled one XCD's work)"""
    share = ((tiles_m + 7) // 8) * tiles_n
    even = (tiles_m * tiles_n + 7) // 8
    return tiles_m >= 8 and (share + 31) // 32 <= (even + 31) // 32


def tile_of(block_id, tiles_m, tiles_n):
    """big_tile, XCD_REMAP branch: (tm, tn) or None for a block beyond its XCD's share"""
    xcd, idx = block_id & 7, block_id >> 3
    if not panels(tiles_m, tiles_n):                       # even runs of the GM-grouped tile list
        nt = tiles_m * tiles_n
        q, r = nt >> 3, nt & 7
        if idx >= q + (1 if xcd < r else 0):
            return None
        lin = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
        group_sz = GM * tiles_n
        first_m = (lin // group_sz) * GM
        gm = min(GM, tiles_m - first_m)
        rin = lin % group_sz
        return first_m + rin % gm, rin // gm
    q, r = tiles_m >> 3, tiles_m & 7
    cnt = q + (1 if xcd < r else 0)
    p0 = xcd * q + min(xcd, r)
    if idx >= cnt * tiles_n:
        return None
    group_sz = GM * tiles_n
    first = (idx // group_sz) * GM
    gm = min(GM, cnt - first)
    rin = idx - (idx // group_sz) * group_sz
    return p0 + first + rin % gm, rin // gm


def grid_of(tiles_m, tiles_n):
    if not panels(tiles_m, tiles_n):
        return 8 * ((tiles_m * tiles_n + 7) // 8)
    return 8 * ((tiles_m + 7) // 8) * tiles_n


@pytest.mark.parametrize('tiles_m', list(range(1, 42)) + [65, 98, 155, 392])
@pytest.mark.parametrize('tiles_n', [1, 2, 3, 6, 18, 24])
def test_every_tile_once_and_panels_stay_on_one_xcd(tiles_m, tiles_n):
    seen, panel_xcd, xcd_panels, idle = {}, {}, {x: [] for x in range(8)}, 0
    for b in range(grid_of(tiles_m, tiles_n)):
        t = tile_of(b, tiles_m, tiles_n)
        if t is None:
            idle += 1
            continue
        tm, tn = t
        assert 0 <= tm < tiles_m and 0 <= tn < tiles_n, (b, t)
        assert t not in seen, f'tile {t} computed by blocks {seen[t]} and {b}'
        seen[t] = b
        if panels(tiles_m, tiles_n):
            assert panel_xcd.setdefault(tm, b & 7) == (b & 7), f'row panel {tm} is split over XCDs'
        if tm not in xcd_panels[b & 7]:
            xcd_panels[b & 7].append(tm)
    assert len(seen) == tiles_m * tiles_n
    assert idle == grid_of(tiles_m, tiles_n) - tiles_m * tiles_n
    # the busiest XCD never needs more rounds of its 32 CUs than an even split of the tile list (the host cost model prices ceil(tiles / 256) rounds)
    per_xcd = [sum(1 for b in seen.values() if b & 7 == x) for x in range(8)]
    assert (max(per_xcd) + 31) // 32 <= ((tiles_m * tiles_n + 7) // 8 + 31) // 32, per_xcd
    if not panels(tiles_m, tiles_n):
        assert max(per_xcd) - min(per_xcd) <= 1
        return
    shares = []
    for x in range(8):
        ps = sorted(xcd_panels[x])
        assert ps == list(range(ps[0], ps[0] + len(ps))) if ps else True          # consecutive panels
        shares.append(len(ps))
    assert max(shares) - min(shares) <= 1 and sum(shares) == tiles_m                # balanced to within one panel
    # blocks that are resident on an XCD at the same time (32 consecutive values of idx) cover at most GM row panels per group of column tiles:
    # within a group the row index runs fastest
    for x in range(8):
        order = [tile_of(i * 8 + x, tiles_m, tiles_n) for i in range(grid_of(tiles_m, tiles_n) // 8)]
        order = [t for t in order if t is not None]
        for i in range(0, len(order), GM * tiles_n):
            grp = order[i:i + GM * tiles_n]
            assert len({tm for tm, _ in grp}) <= GM


def test_grouped_conv_block_offsets_are_multiples_of_eight():
    """host side of the grouped convolution launch (gemm.hip): problem i starts at block tile_start[i], a multiple of 8"""
    problems = [(392, 2), (25, 6), (98, 3), (25, 3), (1, 1), (7, 5)]
    start, starts = 0, []
    for tm, tn in problems:
        starts.append(start)
        start += grid_of(tm, tn)
    assert all(s % 8 == 0 for s in starts)
    # a block of problem i keeps its XCD: (global id) mod 8 == (local id) mod 8
    for s in starts:
        for local in (0, 1, 7, 8, 13):
            assert (s + local) % 8 == local % 8


def test_model_matches_the_sources():
    big = open(os.path.join(SRC, 'gemm_big.hip')).read()
    host = open(os.path.join(SRC, 'gemm.hip')).read()
    assert 'constexpr int GM = 4;' in big
    assert 'const int xcd = block_id & 7, idx = block_id >> 3;' in big
    common = open(os.path.join(SRC, 'gemm_common.h')).read()
    assert 'const int q = p.tiles_m >> 3, r = p.tiles_m & 7;' in big
    assert 'const int cnt = q + (xcd < r ? 1 : 0), p0 = xcd * q + min(xcd, r);' in big
    assert 'if (idx >= cnt * p.tiles_n) return;' in big
    assert 'tm = p0 + first + rin % gm; tn = rin / gm;' in big
    assert 'if (big_xcd_panels(p.tiles_m, p.tiles_n)) {' in big
    assert 'const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;' in big
    assert 'return tiles_m >= 8 && (share + 31) / 32 <= (big_xcd_even_share(tiles_m, tiles_n) + 31) / 32;' in common
    assert 'dim3(big_xcd_grid(p.tiles_m, p.tiles_n))' in big          # single launch: 8 x largest share
    assert 'big_tile<VARIANT, false, false, true, 1>(g.p[i], b - g.tile_start[i]);' in big    # grouped conv: per-problem numbering
    assert 'blocks += big_xcd_grid(g.p[i].tiles_m, g.p[i].tiles_n);' in host
