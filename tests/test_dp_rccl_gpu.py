"""GPU, world_size 2 on TWO devices over RCCL (backend 'nccl' = RCCL on ROCm): the Trainer's data-parallel step with the real
transport, compared with the same step over gloo.  Skipped on the one-GPU development / test boxes; the driver's multi-GPU node
(or any >= 2-GPU box) runs it.  Covers train_caption.py:92-117 (DDP through accelerate) on hardware."""
import os

import pytest
import torch
import torch.multiprocessing as mp

from tests.test_dp_gpu import _free_port, _worker

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL refuses two ranks per device)')
@pytest.mark.parametrize('shard,transport', [(False, 'torch.distributed'), ('rs_ag', 'torch.distributed'), (False, 'native'), ('rs_ag', 'native'), (True, 'native')])
def test_two_ranks_two_gpus_rccl_matches_gloo(shard, transport):
    """transport='native' (round 6): the library's own communicator carries all-reduce, reduce-scatter + all-gather ('rs_ag') and the broadcasts of the
    ZeRO-1 style sharded optimizer (shard_optimizer=True) -- include/prismer_comm.h ph_reduce_scatter / ph_all_gather / ph_broadcast"""
    world = 2
    res = {}
    for backend in ('nccl', 'gloo'):
        mgr = mp.Manager(); out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), True, out, 'fp32', 2, shard, backend, True, transport if backend == 'nccl' else 'torch.distributed'),
                 nprocs=world, join=True)
        res[backend] = (out[0], out[1])
    for backend in res:
        a, b = res[backend]
        for pa, pb in zip(a['params'], b['params']):            # both ranks hold the same parameters after the exchange + update
            assert torch.equal(pa, pb), backend
    for i in range(2):                                          # RCCL and gloo agree up to summation order / atomics noise
        d = (res['nccl'][0]['params'][i] - res['gloo'][0]['params'][i]).abs()
        assert d.max() <= 2.1e-3 * 2 and (d > 1e-5).float().mean() < 5e-2, (i, float(d.max()))
    assert abs(res['nccl'][0]['loss'] - res['gloo'][0]['loss']) < 1e-3 * abs(res['gloo'][0]['loss'])
