"""Model check of the gathered LDS-DMA source addresses of the 256x128 implicit-GEMM convolution kernel (prismer_amd/csrc/gemm_big.hip,
big_tile<..., CONV = 1>, round 4).

The kernel never builds an im2col matrix: every lane of every global_load_lds instruction computes the address of ITS 16-byte chunk of
the logical matrix col[m][k] (m = output pixel, k = (tap, channel)) from loop-invariant per-row state (row_off, tap_ok) and per-k-tile
state (tap_off, tap_bit) -- the latter either per lane with a float-reciprocal division (any C % 8 == 0) or wave-uniform and
incremental on the scalar unit (C % 64 == 0).  A wrong index shows up on hardware only as a small numeric error in a border pixel.
This test re-states that arithmetic in Python, lane by lane, rebuilds the A tile the DMA would deposit in LDS (slot l & 7 of row r
holds chunk (l & 7) ^ ((r >> 1) & 7)) and compares it with a numpy im2col of the same window -- for the forward 3x3 windows (stride 1
and 2), the parity-class windows of the stride-2 data gradient (1..2 x 1..2 taps, offset 0), ragged M, and K that is not a multiple of
the 64-wide k-tile.  The GPU tests (test_implicit_gemm_conv_forward_stats_and_wgrad, test_implicit_conv_data_gradient) check the
implementation; this checks the index algebra it rests on, and pins the source lines that carry it."""
import os

import numpy as np
import pytest

BM, BK, A_INSTR, WAVES = 256, 64, 4, 8
SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'prismer_amd', 'csrc', 'gemm_big.hip')


def fdiv(a, d, inv):
    """gemm_common.h fdiv: a / d through one float multiply and two fix-ups (exact for 0 <= a < 2^24)"""
    q = int(np.float32(np.float32(a) * np.float32(inv)))
    if (q + 1) * d <= a:
        q += 1
    if q * d > a:
        q -= 1
    return q


def pix_of(geo, m):
    B, H, W, C, kh, kw, offy, offx, stride, Ho, Wo = geo
    b = fdiv(m, Ho * Wo, np.float32(1.0) / np.float32(Ho * Wo))
    rem = m - b * Ho * Wo
    oy = fdiv(rem, Wo, np.float32(1.0) / np.float32(Wo))
    ox = rem - oy * Wo
    return b * H * W, oy * stride + offy, ox * stride + offx


def im2col(x, geo, K):
    B, H, W, C, kh, kw, offy, offx, stride, Ho, Wo = geo
    col = np.zeros((B * Ho * Wo, K), dtype=x.dtype)
    for b in range(B):
        for oy in range(Ho):
            for ox in range(Wo):
                m = (b * Ho + oy) * Wo + ox
                for ty in range(kh):
                    for tx in range(kw):
                        iy, ix = oy * stride + offy + ty, ox * stride + offx + tx
                        if 0 <= iy < H and 0 <= ix < W:
                            col[m, (ty * kw + tx) * C:(ty * kw + tx + 1) * C] = x[b, iy, ix]
    return col


def gathered_tile(x, geo, K, m0, incremental):
    """the [256][nk*64] A tile as the kernel's lanes fetch it (None where the zero page is read)"""
    B, H, W, C, kh, kw, offy, offx, stride, Ho, Wo = geo
    Kreal = kh * kw * C
    M = B * Ho * Wo
    nk = (K + BK - 1) // BK
    xf = x.reshape(-1)
    inv_c = np.float32(1.0) / np.float32(C)
    tile = np.zeros((BM, nk * BK), dtype=x.dtype)
    written = np.zeros((BM, nk * BK // 8), dtype=np.int32)
    for wave in range(WAVES):
        for lane in range(64):
            conv_k = ((lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7)) * 8
            rows, row_off, tap_ok = [], [], []
            for i in range(A_INSTR):
                r = (i * 8 + wave) * 8 + (lane >> 3)
                assert ((lane & 7) ^ ((r >> 1) & 7)) * 8 == conv_k          # the chunk does not depend on the instruction index
                base, iy0, ix0 = pix_of(geo, min(m0 + r, M - 1))
                rows.append(r)
                row_off.append((base + iy0 * W + ix0) * C)
                m = 0
                for ty in range(kh):
                    for tx in range(kw):
                        if 0 <= iy0 + ty < H and 0 <= ix0 + tx < W:
                            m |= 1 << (ty * kw + tx)
                tap_ok.append(m)
            s_ky = s_kx = s_c0 = s_tap = 0
            for kt in range(nk):
                k = kt * BK + conv_k
                if incremental:
                    tap_off = (s_ky * W + s_kx) * C + s_c0 + conv_k
                    tap_bit = (1 << s_tap) if s_tap < 16 else 0
                    s_c0 += BK
                    if s_c0 >= C:
                        s_c0 = 0; s_tap += 1; s_kx += 1
                        if s_kx == kw:
                            s_kx = 0; s_ky += 1
                else:
                    kin = k < Kreal
                    kk = k if kin else 0
                    tap = fdiv(kk, C, inv_c)
                    c0 = kk - tap * C
                    ky = (2 if tap >= 6 else (1 if tap >= 3 else 0)) if kw == 3 else ((tap >> 1) if kw == 2 else tap)
                    kx = tap - ky * kw
                    tap_off = (ky * W + kx) * C + c0
                    tap_bit = (1 << tap) if kin else 0
                for i in range(A_INSTR):
                    chunk = conv_k // 8
                    written[rows[i], kt * 8 + chunk] += 1
                    if tap_ok[i] & tap_bit:
                        off = row_off[i] + tap_off
                        assert 0 <= off and off + 8 <= xf.size, (off, xf.size)           # a selected address is always inside the activation
                        tile[rows[i], k:k + 8] = xf[off:off + 8]
    assert (written == 1).all()                                                       # every 16-B chunk of the tile is fetched exactly once
    return tile


FWD = lambda B, H, C, s: (B, H, H, C, 3, 3, -1, -1, s, (H + 2 - 3) // s + 1, (H + 2 - 3) // s + 1)
DGRAD = lambda B, Ho, C, py, px: (B, Ho, Ho, C, 1 + py, 1 + px, 0, 0, 1, Ho, Ho)


@pytest.mark.parametrize('geo,K,m0,incremental', [
    (FWD(2, 12, 64, 1), 9 * 64, 0, True), (FWD(2, 12, 64, 1), 9 * 64, 0, False), (FWD(2, 12, 64, 1), 9 * 64, 256, True),      # ragged last tile (M = 288)
    (FWD(3, 14, 128, 2), 9 * 128, 0, True), (FWD(3, 14, 128, 2), 9 * 128, 0, False),
    (FWD(2, 16, 40, 2), 9 * 40, 0, False), (FWD(3, 10, 24, 1), 216, 256, False),          # C < 64, K = 360 / 216: not a multiple of 64
    (FWD(2, 15, 96, 2), 864, 0, False),                                                   # the 96-channel stem layer: a k-tile straddles taps
    (DGRAD(2, 12, 64, 1, 1), 4 * 64, 0, True), (DGRAD(2, 12, 64, 0, 1), 2 * 64, 0, True), (DGRAD(2, 12, 64, 1, 0), 2 * 64, 0, False),
    (DGRAD(3, 10, 192, 0, 0), 192, 0, True),
])
def test_gathered_addresses_rebuild_the_im2col_tile(geo, K, m0, incremental):
    B, H, W, C = geo[:4]
    rng = np.random.default_rng(5)
    x = rng.integers(1, 2 ** 15, size=(B, H, W, C)).astype(np.int32)              # nonzero everywhere: a missed element cannot hide as padding
    col = im2col(x, geo, K)
    M = col.shape[0]
    tile = gathered_tile(x, geo, K, m0, incremental)
    rows = min(BM, M - m0)
    assert np.array_equal(tile[:rows, :K], col[m0:m0 + rows])
    assert not tile[:rows, K:].any()                                               # chunks beyond the real K read the zero page
    if rows < BM:                                                                  # rows beyond M re-read the last valid pixel (masked by the write-out)
        assert np.array_equal(tile[rows:, :K], np.broadcast_to(col[M - 1], (BM - rows, K)))


def test_incremental_state_past_the_last_tap_reads_the_zero_page():
    """the non-LEAN loop re-requests its last tile with a clamped index: the wave-uniform state has then run PAST the window (tap >= kh*kw)
    and must select the zero page, never an address"""
    for kh, kw in ((3, 3), (1, 1), (2, 1), (1, 2), (2, 2)):
        full = (1 << (kh * kw)) - 1                       # tap_ok of an interior pixel
        for extra in range(1, 4):
            s_tap = kh * kw - 1 + extra
            tap_bit = (1 << s_tap) if s_tap < 16 else 0
            assert full & tap_bit == 0


def test_model_matches_the_kernel_source():
    src = open(SRC).read()
    assert 'const int conv_k = (((lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7))) * 8;' in src
    assert 'row_off[i] = (px.base + px.iy0 * p.cv.W + px.ix0) * p.cv.C;' in src
    assert 'm |= ((unsigned)(px.iy0 + ty) < (unsigned)p.cv.H && (unsigned)(px.ix0 + tx) < (unsigned)p.cv.W) ? (1 << (ty * p.cv.kw + tx)) : 0;' in src
    assert 'tap_off = (s_ky * cvW + s_kx) * cvC + s_c0 + conv_k;' in src and 'tap_bit = s_tap < 16 ? (1 << s_tap) : 0;' in src
    assert 'if (s_c0 >= cvC) { s_c0 = 0; ++s_tap; ++s_kx; if (s_kx == cvkw) { s_kx = 0; ++s_ky; } }' in src
    assert 'const bool conv_c64 = CONV == 1 && (cvC & 63) == 0 && (cvKreal & 63) == 0;' in src
    assert 'a_nxt[i] = (tap_ok[i] & tap_bit) ? src : zero_pg;' in src and 'const bf16* src = conv_x + (row_off[i] + tap_off);' in src
    assert 'b_nxt[i] = bin ? src : zero_pg;' in src and 'const bool bin = k < Kfull;' in src
    # every wave asks for its tiles in order (the incremental state depends on it): prologue 0, 1, (2), then t + 2 + grp
    assert 'conv_prep(0);' in src and 'conv_prep(min(1, nk - 1));' in src and 'conv_prep(min(2, nk - 1));' in src
    assert 'if (ISS1 || !grp) conv_prep(min(t + 2 + grp, nk - 1));' in src
