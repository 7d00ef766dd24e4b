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


SITE_STREAM = {'self_probs': 1, 'self_out': 2, 'cross_probs': 3, 'cross_out': 4, 'mlp_out': 5}     # programs/decoder.py: li * 16 + k; embeddings: 9000


class LibraryDropout:
    """`drop(site, x)` for oracle.text_decoder / the patched reference modules with EXACTLY the masks libprismer_hip draws for `seed`:
    hidden-state sites use the linear scheme (element i of the [B*T, H] matrix -> word i % 4 of philox(i / 4, stream)), attention-probability
    sites the (row = (b * heads + h) * Sq + q, key / 4) scheme; stream = layer * 16 + k (programs/decoder.py), 9000 for the embeddings.
    Scaling 1 / (1 - p) with p rounded like the kernels do (threshold = int(p * 2^24))."""

    def __init__(self, seed, p_hidden, p_attn, site_base=0):
        self.seed, self.p_hidden, self.p_attn, self.site_base = int(seed) & 0xFFFFFFFFFFFFFFFF, float(p_hidden), float(p_attn), site_base
        self.calls = []

    def __call__(self, site, x):
        self.calls.append(site)
        if site == ('emb',):
            stream, p, attn = 9000, self.p_hidden, False
        else:
            layer, kind = site
            stream, attn = layer * 16 + SITE_STREAM[kind], kind.endswith('_probs')
            p = self.p_attn if attn else self.p_hidden
        if p <= 0.0:
            return x
        stream += self.site_base
        if attn:
            B, H, Sq, Sk = x.shape
            keep = dropout_keep_attention(B * H, Sq, Sk, self.seed, stream, p).reshape(B, H, Sq, Sk)
        else:
            keep = dropout_keep_linear(x.numel(), self.seed, stream, p).reshape(x.shape)
        return x * keep.to(x.dtype) / (1.0 - p)


def splitmix64(z):
    """csrc/optim.hip advance_seed_kernel: the dropout seed of step k+1 from the seed of step k"""
    m = 0xFFFFFFFFFFFFFFFF
    z = (z + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)
