"""Host-side bookkeeping of the native training step that needs no GPU: which weight gradients may be OVERWRITTEN by their GEMM
(Trainer._finish_count, the rule behind Trainer(overwrite_single_writer=True)) and which chunks AdamW may leave un-zeroed (ph_adamw_keep bitmap)."""
import types

import torch

from prismer_amd.trainer import Trainer


class _Store:
    def __init__(self, n, n_train):
        self.grad = torch.zeros(n)
        self.n_train = n_train


def _run(counts_by_offset, n=10000, n_train=9000):
    st = _Store(n, n_train)
    base = st.grad.data_ptr()
    fake = types.SimpleNamespace(stores=[st], device='cpu')
    counts = {base + 4 * off: v for off, v in counts_by_offset.items()}
    Trainer._finish_count(fake, counts)
    bits = fake._keep_maps[0].numpy().view('uint32')
    kept = [c for c in range((n_train + 1023) // 1024) if (bits[c >> 5] >> (c & 31)) & 1]
    return {(k - base) // 4 for k in fake._exclusive}, kept, fake._excl_ranges[0]


def test_single_contiguous_writers_are_exclusive_and_only_whole_chunks_are_kept():
    # (writes per step, numel, contiguous & deferred)
    excl, kept, ranges = _run({0: (1, 3000, True), 3000: (1, 2500, True)})
    assert excl == {0, 3000} and sorted(ranges) == [(0, 3000), (3000, 2500)]
    # chunk c = gradients [1024 c, 1024 c + 1024): 0, 1 lie inside [0, 3000); 2 straddles the boundary; 3, 4 inside [3000, 5500) -> [3072, 5120)
    assert kept == [0, 1, 3, 4]


def test_two_writers_strided_outputs_frozen_tail_and_overlaps_keep_accumulating():
    excl, kept, _ = _run({
        0: (2, 2048, True),          # tied / micro-batched: written twice per step
        2048: (1, 2048, False),      # strided view or a launch that bypassed the queue
        4096: (1, 2048, True),       # fine
        6144: (1, 2048, True),       # overlaps the next one
        7168: (1, 1024, True),
        8192: (1, 1024, True),       # reaches beyond the trainable range (n_train = 9000)
    })
    assert excl == {4096}
    assert kept == [4, 5]


def test_outputs_outside_the_gradient_buffer_are_ignored():
    st = _Store(4096, 4096)
    other = torch.zeros(4096)
    fake = types.SimpleNamespace(stores=[st], device='cpu')
    Trainer._finish_count(fake, {other.data_ptr(): (1, 1024, True), st.grad.data_ptr() + 2: (1, 8, True)})     # foreign buffer; misaligned
    assert fake._exclusive == set() and int(fake._keep_maps[0].abs().sum()) == 0


def test_bench_loss_check_rejects_garbage():
    """bench.check_loss (round 5): a leg whose loss is not a sane training loss raises instead of publishing a throughput (the round-4 loader leg
    reported 1.7e8 and 1.1e18)"""
    import math
    import pytest
    import bench
    bench.check_loss('ok', 283.8, 260.4)
    bench.check_loss('ok', 32.6, 23.2)
    for first, last in ((283.8, 1.7e8), (283.8, 1.147e18), (283.8, float('nan')), (283.8, float('inf')), (283.8, -1.19e34), (283.8, 0.0), (float('nan'), 260.0),
                        (283.8, 600.0)):
        with pytest.raises(RuntimeError, match='loss check failed'):
            bench.check_loss('leg', first, last)
    assert math.isfinite(283.8)
