"""Gradient exchange of the data-parallel training step (reference: torch DDP through accelerate, train_caption.py:92-93,117:
gradients are averaged over ranks every step).

Each rank owns flat fp32 gradient buffers (prismer_amd/store.py).  The Trainer cuts its backward into segments in reverse
layer order; as soon as a segment has finished, the buffer ranges whose gradients are complete are handed to
`GradExchange.issue`, which -- on the communication stream, behind an event recorded after that segment -- packs the range
to bf16 (half the bytes on the xGMI links: 485 MB instead of 970 MB per step for Prismer-BASE), SUM-all-reduces it in chunks
and unpacks it back into the fp32 buffer, all overlapping the remaining backward.  The 1/world factor is folded into the
fused AdamW (`grad_scale`).  `payload='fp32'` skips the pack/unpack (bit-for-bit DDP semantics).

Transport: RCCL, reached either through torch.distributed (backend 'nccl' = RCCL; default) or through the library's own
communicator (include/prismer_comm.h, `transport='native'`).  Nothing here needs a GPU by itself: the pack / unpack / reduce
callables are injected, which is how tests/test_dist_cpu.py runs the same choreography over gloo on CPU tensors.
"""
import torch
import torch.distributed as dist


def bucket_ranges(n, bucket_elems, start=0):
    return [(o, min(n, o + bucket_elems)) for o in range(start, n, bucket_elems)]


def bucketed_all_reduce(flat, n, bucket_elems, group=None, async_op=False):
    """plain fp32 SUM all-reduce of flat[:n] in fixed-size buckets (round-1 path; kept for the fp32 payload and the tests)"""
    works = []
    for lo, hi in bucket_ranges(n, bucket_elems):
        w = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def broadcast_flat(flat, src=0, group=None):
    dist.broadcast(flat, src, group=group)


def contiguous_stages(names, offset, numel, stage_of, align):
    """[(stage, lo, hi)]: maximal runs of consecutive parameters (buffer order) that share a completion stage.  `hi` is
    rounded up to the store's alignment so that consecutive runs tile the buffer without gaps."""
    runs = []
    for n in names:
        s = stage_of(n)
        lo = offset[n]
        hi = lo + (numel[n] + align - 1) // align * align
        if runs and runs[-1][0] == s and runs[-1][2] == lo:
            runs[-1][2] = hi
        else:
            runs.append([s, lo, hi])
    return [tuple(r) for r in runs]


class GradExchange:
    def __init__(self, world, all_reduce, pack=None, unpack=None, payload='bf16', chunk_elems=64 << 20, comm_stream=None,
                 transport='torch.distributed'):
        """all_reduce(t): in-place SUM over ranks, enqueued on the CURRENT stream (torch: dist.all_reduce; native:
        ph_allreduce_bucket).  pack(src_f32, dst_bf16) / unpack(src_bf16, dst_f32): cast kernels (bf16 payload only)."""
        assert payload in ('bf16', 'fp32')
        self.world, self.all_reduce, self.pack, self.unpack = world, all_reduce, pack, unpack
        self.payload, self.chunk, self.comm_stream, self.transport = payload, chunk_elems, comm_stream, transport
        self.scratch = {}          # id(flat) -> bf16 bucket buffer of the same length
        self.log = []              # [(tag, lo, hi, n_collectives)] in issue order (tests assert the overlap structure on this)
        self.bytes_per_step = 0

    def describe(self):
        return dict(payload=self.payload, transport=self.transport, chunk_mb=self.chunk * (2 if self.payload == 'bf16' else 4) >> 20,
                    collectives_per_step=sum(e[3] for e in self.log_last), bytes_per_step=self.bytes_last,
                    ranges_per_step=len(self.log_last))

    log_last, bytes_last = (), 0

    def begin_step(self):
        if self.log:
            self.log_last, self.bytes_last = tuple(self.log), self.bytes_per_step
        self.log, self.bytes_per_step = [], 0

    def _bucket(self, flat):
        b = self.scratch.get(id(flat))
        if b is None:
            b = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            self.scratch[id(flat)] = b
        return b

    def issue(self, flat, lo, hi, tag='', after=None):
        """all-reduce flat[lo:hi] on the communication stream once `after` (an event on the compute stream) has fired."""
        if self.world == 1 or hi <= lo:
            return
        ctx = torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else _null()
        if after is not None and self.comm_stream is not None:
            self.comm_stream.wait_event(after)
        n = 0
        with ctx:
            for a, b in bucket_ranges(hi, self.chunk, lo):
                if self.payload == 'bf16':
                    buf = self._bucket(flat)
                    self.pack(flat[a:b], buf[a:b])
                    self.all_reduce(buf[a:b])
                    self.unpack(buf[a:b], flat[a:b])
                    self.bytes_per_step += 2 * (b - a)
                else:
                    self.all_reduce(flat[a:b])
                    self.bytes_per_step += 4 * (b - a)
                n += 1
        self.log.append((tag, lo, hi, n))


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
