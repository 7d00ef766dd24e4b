"""CPU, world_size 2, gloo: the N>1 gradient exchange (bucketed SUM all-reduce of a flat buffer + 1/world scaling in the
optimizer) is equivalent to one big-batch step -- the DDP equivalence of SURVEY section 4."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from prismer_amd.dist import bucket_ranges, bucketed_all_reduce


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    w = torch.randn(1000)                          # identical parameters on both ranks
    torch.manual_seed(100 + rank)
    x = torch.randn(8, 1000)                       # rank-local batch
    # per-rank gradient of mean_b (x_b . w)^2 over the LOCAL batch
    g = (2 * (x @ w)[:, None] * x).mean(0)
    flat = torch.zeros(1024); flat[:1000] = g
    works = bucketed_all_reduce(flat, 1000, 300, async_op=True)   # 4 buckets, ragged tail
    for wk in works:
        wk.wait()
    out[rank] = (flat[:1000] / world).clone()      # grad_scale = 1/world (Trainer folds it into AdamW)
    dist.destroy_process_group()


def test_bucketed_all_reduce_equals_big_batch():
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    torch.manual_seed(0); w = torch.randn(1000)
    xs = []
    for r in range(world):
        torch.manual_seed(100 + r); xs.append(torch.randn(8, 1000))
    x = torch.cat(xs)
    ref = (2 * (x @ w)[:, None] * x).mean(0)       # one rank with the 2x batch
    assert torch.allclose(out[0], ref, rtol=1e-5, atol=1e-5) and torch.allclose(out[0], out[1])


def test_bucket_ranges_cover_exactly():
    r = bucket_ranges(1000, 300)
    assert r == [(0, 300), (300, 600), (600, 900), (900, 1000)]
    assert bucket_ranges(5, 16) == [(0, 5)]
