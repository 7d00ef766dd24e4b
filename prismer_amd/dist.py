"""Gradient-exchange helpers that do not depend on a GPU (tested with gloo, world_size 2, on CPU).

Data-parallel semantics of the reference = torch DDP through accelerate (train_caption.py:92-117): gradients are averaged over
ranks every step.  Here each rank owns flat fp32 gradient buffers; `bucketed_all_reduce` sums them in fixed-size buckets
(one collective per bucket, issued in buffer order so that all ranks agree on the sequence) and the 1/world factor is folded
into the optimizer (`grad_scale`)."""
import torch
import torch.distributed as dist


def bucket_ranges(n, bucket_elems):
    return [(o, min(n, o + bucket_elems)) for o in range(0, n, bucket_elems)]


def bucketed_all_reduce(flat, n, bucket_elems, group=None, async_op=False):
    works = []
    for lo, hi in bucket_ranges(n, bucket_elems):
        w = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def broadcast_flat(flat, src=0, group=None):
    dist.broadcast(flat, src, group=group)
