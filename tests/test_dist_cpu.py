"""CPU, world_size 2, gloo: the N>1 gradient exchange (bucketed SUM all-reduce of a flat buffer + 1/world scaling in the
optimizer) is equivalent to one big-batch step -- the DDP equivalence of SURVEY section 4."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from prismer_amd.dist import GradExchange, bucket_ranges, bucketed_all_reduce, contiguous_stages


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


def _worker_bf16(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    g = torch.randn(5000) * torch.logspace(-4, 0, 5000)           # gradients spanning four decades
    flat32, flat16 = g.clone(), g.clone()
    ex32 = GradExchange(world, lambda t: dist.all_reduce(t), payload='fp32', chunk_elems=1024)
    ex16 = GradExchange(world, lambda t: dist.all_reduce(t), pack=lambda s, d, scale=1.0: d.copy_(s * scale), unpack=lambda s, d: d.copy_(s),
                        payload='bf16', chunk_elems=1024)
    for ex, flat in ((ex32, flat32), (ex16, flat16)):
        ex.begin_step()
        # stages arrive in REVERSE buffer order, like the backward: tail first, head last; a stage may own several ranges
        ex.issue(flat, 3000, 5000, 'late')
        ex.issue(flat, 1000, 3000, 'mid')
        ex.issue(flat, 0, 1000, 'early')
    out[rank] = (flat32, flat16, list(ex16.log), ex16.bytes_per_step, ex32.bytes_per_step)
    dist.destroy_process_group()


def test_bf16_bucket_exchange_matches_fp32_within_1e2():
    """bf16 payload (pack -> all-reduce -> unpack per chunk, issued per finished backward stage) vs the fp32 exchange."""
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker_bf16, args=(world, _free_port(), out), nprocs=world, join=True)
    f32, f16, log, b16, b32 = out[0]
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])       # every rank ends with the same sums
    # (round 3) the bf16 payload is pre-scaled by 1/world before the rounding: it delivers the MEAN (post_scale 1), the fp32 payload the SUM
    f16 = f16 * world
    rel = ((f16 - f32).norm() / f32.norm()).item()
    assert rel < 1e-2, rel
    mag = torch.zeros(5000)
    for r in range(world):
        torch.manual_seed(100 + r)
        mag += (torch.randn(5000) * torch.logspace(-4, 0, 5000)).abs()
    assert ((f16 - f32).abs() <= 2.0 ** -7 * mag + 1e-9).all()                             # element-wise: bf16 rounding of each operand + of the sum
    assert [e[0] for e in log] == ['late', 'mid', 'early'] and [e[3] for e in log] == [2, 2, 1]
    assert b16 * 2 == b32 == 4 * 5000


def test_contiguous_stages_cover_the_buffer_in_order():
    names = ['emb', 'l0.a', 'l0.kv', 'l1.a', 'l2.a', 'head']
    numel = {'emb': 100, 'l0.a': 64, 'l0.kv': 300, 'l1.a': 64, 'l2.a': 70, 'head': 10}
    offset, o = {}, 0
    for n in names:
        offset[n] = o; o += (numel[n] + 63) // 64 * 64
    stage = {'emb': 'last', 'l0.a': 'last', 'l0.kv': 'last', 'l1.a': 'mid', 'l2.a': 'first', 'head': 'first'}
    runs = contiguous_stages(names, offset, numel, stage.get, 64)
    assert runs == [('last', 0, 128 + 64 + 320), ('mid', 512, 576), ('first', 576, 576 + 128 + 64)]
    assert runs[0][1] == 0 and all(a[2] == b[1] for a, b in zip(runs, runs[1:]))


def _worker_rs(rank, world, port, out, payload):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    g = torch.randn(5003) * torch.logspace(-3, 0, 5003)
    flat = g.clone()
    ex = GradExchange(world, lambda t: dist.all_reduce(t), pack=lambda s, d, scale=1.0: d.copy_(s * scale), unpack=lambda s, d: d.copy_(s),
                      payload=payload, mode='rs_ag', rank=rank,
                      reduce_scatter=lambda o, i: dist.reduce_scatter_tensor(o, i), all_gather=lambda o, i: dist.all_gather_into_tensor(o, i))
    ex.begin_step()
    ex.issue(flat, 3000, 5003, 'late')            # ragged: 2003 elements over `world` chunks of 256-element granularity
    ex.issue(flat, 700, 3000, 'mid')
    ex.issue(flat, 0, 700, 'early')               # (a range shorter than world * 256: the last ranks own nothing)
    owned = dict(ex.owned[id(flat)])
    # "optimizer": the owner turns its reduced gradient piece into new parameter values; everybody else's copy is stale
    params = torch.full((5003,), float('nan'))
    for lo, (a, b) in owned.items():
        params[a:b] = flat[a:b] * ex.post_scale * 2.0 + 1.0
    ex.gather(flat, params)
    out[rank] = (flat.clone(), params.clone(), owned, ex.post_scale)
    dist.destroy_process_group()


@pytest.mark.parametrize('world,payload', [(2, 'fp32'), (3, 'fp32'), (2, 'bf16')])
def test_reduce_scatter_all_gather_exchange(world, payload):
    """mode='rs_ag': every rank ends up owning disjoint pieces that tile each issued range and hold the reduced gradient there; the
    all-gather of values computed on the owned pieces reconstructs the same full vector on every rank."""
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker_rs, args=(world, _free_port(), out, payload), nprocs=world, join=True)
    gs = []
    for r in range(world):
        torch.manual_seed(100 + r); gs.append(torch.randn(5003) * torch.logspace(-3, 0, 5003))
    mean = sum(gs) / world
    cover = torch.zeros(5003)
    for r in range(world):
        flat, params, owned, post = out[r]
        for lo, (a, b) in owned.items():
            cover[a:b] += 1
            got = flat[a:b] * post                                               # the average DDP would hand the optimizer
            tol = 1e-6 if payload == 'fp32' else 2.0 ** -7
            assert (got - mean[a:b]).abs().max() <= tol * sum(x[a:b].abs() for x in gs).max() + 1e-9
    assert torch.equal(cover, torch.ones(5003))                                  # the owned pieces tile [0, 5003) exactly once
    ref = out[0][1]
    assert torch.isfinite(ref).all()
    for r in range(1, world):
        assert torch.equal(out[r][1], ref)                                       # identical parameters everywhere after the gather
    want = mean * 2.0 + 1.0
    assert (ref - want).abs().max() < (1e-5 if payload == 'fp32' else 5e-2)


def test_exchange_busy_time_accounting_on_a_fake_clock():
    """GradExchange.timing brackets every issue() with two events and sums their distances per step (bench.py's comm_ms_total).  Here the
    events read a fake clock that the injected collectives advance by known amounts: three steps, three stages each, buckets included."""
    from prismer_amd.dist import GradExchange
    clock = [0.0]

    class Ev:
        def record(self):
            self.t = clock[0]

        def elapsed_time(self, other):
            return other.t - self.t
    delays = []

    def all_reduce(t):
        d = 0.25 + 1e-6 * t.numel()            # ms: a latency term plus a bandwidth term
        delays.append(d)
        clock[0] += d
    ex = GradExchange(4, all_reduce, payload='fp32', chunk_elems=1000, event_factory=Ev)
    ex.timing = True
    flat = torch.zeros(5000)
    want = []
    for step in range(3):
        ex.begin_step()
        n0 = len(delays)
        clock[0] += 3.0                          # compute between the stages: must NOT be counted as communication
        ex.issue(flat, 0, 2500, 'dec0')          # 3 buckets of <= 1000 elements
        clock[0] += 1.5
        ex.issue(flat, 2500, 2600, 'trunk')      # 1 bucket
        ex.issue(flat, 2600, 5000, 'front')      # 3 buckets
        want.append(sum(delays[n0:]))
        assert [e[3] for e in ex.log] == [3, 1, 3]
    got = ex.collect_timing()
    assert len(got) == 3 and all(abs(g - w) < 1e-9 for g, w in zip(got, want)), (got, want)
    assert ex.bytes_per_step == 4 * 5000


def test_exchange_timeline_predictor():
    """predict_exchange (the model behind DESIGN section 6's 8-GPU expectation): hidden, exposed and serialised cases"""
    from prismer_amd.dist import predict_exchange
    one = predict_exchange([10.0, 5.0], [4e8, 1e8], 100.0, 1)
    assert one['comm_ms_total'] == 0.0 and one['comm_ms_exposed'] == 0.0 and one['step_ms'] == 15.0
    # 8 ranks, 100 GB/s per link: 400 MB all-reduce = 2 * 7/8 * 4 ms = 7 ms (+ latency), hidden behind the 10 ms that follow it
    hid = predict_exchange([5.0, 10.0], [4e8, 0], 100.0, 8, latency_us=0.0)
    assert abs(hid['comm_ms_total'] - 7.0) < 1e-9 and hid['comm_ms_exposed'] == 0.0 and hid['step_ms'] == 15.0
    # the same payload completed by the LAST segment is fully exposed
    exp = predict_exchange([5.0, 10.0], [0, 4e8], 100.0, 8, latency_us=0.0)
    assert abs(exp['comm_ms_exposed'] - 7.0) < 1e-9 and abs(exp['step_ms'] - 22.0) < 1e-9
    # a stage cannot start before the previous one is through: 7 ms + 7 ms behind 5 + 2 ms of compute -> 12 ms exposed... minus the overlap
    ser = predict_exchange([5.0, 2.0], [4e8, 4e8], 100.0, 8, latency_us=0.0)
    assert abs(ser['comm_ms_total'] - 14.0) < 1e-9 and abs(ser['comm_ms_exposed'] - 12.0) < 1e-9
    rs = predict_exchange([5.0], [4e8], 100.0, 8, latency_us=0.0, mode='rs_ag')
    assert abs(rs['comm_ms_total'] - 3.5) < 1e-9

