"""Test helpers: error metrics and a numpy Philox4x32-7 mirror of csrc/common.h (dropout masks)."""
import numpy as np
import torch


def rel_fro(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_abs(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _mulhilo(a, b):
    p = a.astype(np.uint64) * np.uint64(b)
    return (p >> np.uint64(32)).astype(np.uint32), (p & np.uint64(0xFFFFFFFF)).astype(np.uint32)


PHILOX_ROUNDS = 7


def philox4x32(c0, c1, c2, c3, k0, k1):
    """vectorised Philox4x32 with PHILOX_ROUNDS rounds (= PH_PHILOX_ROUNDS of csrc/common.h); counters are numpy uint32 arrays, keys python ints.
    Returns 4 uint32 arrays."""
    c0 = c0.astype(np.uint32); c1 = np.broadcast_to(np.uint32(c1), c0.shape).copy() if np.isscalar(c1) else c1.astype(np.uint32)
    c2 = np.broadcast_to(np.uint32(c2), c0.shape).copy() if np.isscalar(c2) else c2.astype(np.uint32)
    c3 = np.broadcast_to(np.uint32(c3), c0.shape).copy() if np.isscalar(c3) else c3.astype(np.uint32)
    k0 = np.uint32(k0 & 0xFFFFFFFF); k1 = np.uint32(k1 & 0xFFFFFFFF)
    for _ in range(PHILOX_ROUNDS):
        hi0, lo0 = _mulhilo(c0, 0xD2511F53)
        hi1, lo1 = _mulhilo(c2, 0xCD9E8D57)
        n0 = hi1 ^ c1 ^ k0; n1 = lo1; n2 = hi0 ^ c3 ^ k1; n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = np.uint32((int(k0) + 0x9E3779B9) & 0xFFFFFFFF); k1 = np.uint32((int(k1) + 0xBB67AE85) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def dropout_keep_linear(n_elems, seed, stream, p):
    """keep mask (bool [n_elems]) for the GEMM-epilogue / LayerNorm / embedding dropout: element i uses word i%4 of
    philox(counter=(i/4 lo, i/4 hi, stream, 0x5eed), key=seed)."""
    assert n_elems % 4 == 0
    idx4 = np.arange(n_elems // 4, dtype=np.uint64)
    r = philox4x32((idx4 & 0xFFFFFFFF).astype(np.uint32), (idx4 >> 32).astype(np.uint32), stream, 0x5eed,
                   seed & 0xFFFFFFFF, seed >> 32)
    r = np.stack(r, axis=1).reshape(-1)
    thr = np.uint32(int(np.float32(p) * np.float32(16777216.0)))
    return torch.from_numpy((r >> np.uint32(8)) >= thr)


def dropout_keep_attention(BH, Sq, Sk, seed, stream, p):
    """keep mask [BH, Sq, Sk] for attention-probability dropout: philox(counter=(key/4, row id, stream, 0xa77e))."""
    k4 = (Sk + 3) // 4
    rows = np.arange(BH * Sq, dtype=np.uint32)
    c0 = np.tile(np.arange(k4, dtype=np.uint32), BH * Sq)
    c1 = np.repeat(rows, k4)
    r = philox4x32(c0, c1, stream, 0xa77e, seed & 0xFFFFFFFF, seed >> 32)
    r = np.stack(r, axis=1).reshape(BH * Sq, k4 * 4)[:, :Sk]
    thr = np.uint32(int(np.float32(p) * np.float32(16777216.0)))
    return torch.from_numpy((r >> np.uint32(8)) >= thr).reshape(BH, Sq, Sk)
