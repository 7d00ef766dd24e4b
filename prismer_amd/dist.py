"""Gradient exchange of the data-parallel training step (reference: torch DDP through accelerate, train_caption.py:92-93,117:
gradients are averaged over ranks every step).

Each rank owns flat fp32 gradient buffers (prismer_amd/store.py).  The Trainer cuts its backward into segments in reverse
layer order; as soon as a segment has finished, the buffer ranges whose gradients are complete are handed to
`GradExchange.issue`, which -- on the communication stream, behind an event recorded after that segment -- packs the range
to bf16 (half the bytes on the xGMI links: 485 MB instead of 970 MB per step for Prismer-BASE), SUM-all-reduces it in chunks
and unpacks it back into the fp32 buffer, all overlapping the remaining backward.

Payload (round 3): the DEFAULT is 'fp32' -- DDP's own payload (SUM of fp32 gradients, 1/world folded into the fused AdamW as
`post_scale`).  'bf16' is opt-in: every rank pre-scales its gradient by 1/world BEFORE the cast (the sum of world bf16 values
then has the magnitude of one of them: no overflow head-room lost, no extra rounding step after the sum) and `post_scale` is 1.

Mode (round 3): 'allreduce' (above) or 'rs_ag' -- reduce-scatter + all-gather, the all-links form SURVEY 8e asks for and the
communication pattern of accelerate's FSDP SHARD_GRAD_OP flag (train_caption.py:56-66): each issued range is cut into `world`
equal chunks, rank r receives the reduced chunk r (reduce-scatter: half the bytes of an all-reduce on every link), runs AdamW on
the chunks it owns (optimizer state and update time 1/world) and the updated fp32 parameters are all-gathered per range.

Transport: RCCL, reached either through torch.distributed (backend 'nccl' = RCCL; default) or through the library's own
communicator (include/prismer_comm.h, `transport='native'`).  Nothing here needs a GPU by itself: the pack / unpack / reduce
callables are injected, which is how tests/test_dist_cpu.py runs the same choreography over gloo on CPU tensors.
"""
import torch
import torch.distributed as dist


def _copy_flat(dst, src):
    """staging copies of the reduce-scatter / all-gather mode: the library's copy kernel on the GPU (round-4 review: no stock torch kernel on the
    step path), Tensor.copy_ for the CPU tensors of the gloo tests"""
    if dst.is_cuda:
        from . import ops
        return ops.copy_flat(dst, src)
    return dst.copy_(src)


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


RS_ALIGN = 256          # chunk granularity of the reduce-scatter (elements): keeps every owned piece 1 KB aligned for the AdamW kernel


def rs_chunk(n, world):
    """elements per rank when a range of n elements is reduce-scattered over `world` ranks"""
    return ((n + world - 1) // world + RS_ALIGN - 1) // RS_ALIGN * RS_ALIGN


class GradExchange:
    def __init__(self, world, all_reduce, pack=None, unpack=None, payload='fp32', chunk_elems=64 << 20, comm_stream=None,
                 transport='torch.distributed', mode='allreduce', rank=0, reduce_scatter=None, all_gather=None, event_factory=None):
        """all_reduce(t): in-place SUM over ranks, enqueued on the CURRENT stream (torch: dist.all_reduce; native:
        ph_allreduce_bucket).  pack(src_f32, dst_bf16, scale) / unpack(src_bf16, dst_f32): cast kernels (bf16 payload only).
        mode 'rs_ag': reduce_scatter(out[c], inp[world * c]) and all_gather(out[world * c], inp[c]) (torch:
        dist.reduce_scatter_tensor / all_gather_into_tensor)."""
        assert payload in ('bf16', 'fp32') and mode in ('allreduce', 'rs_ag')
        self.world, self.all_reduce, self.pack, self.unpack = world, all_reduce, pack, unpack
        self.payload, self.chunk, self.comm_stream, self.transport = payload, chunk_elems, comm_stream, transport
        self.mode, self.rank, self.reduce_scatter, self.all_gather = mode, rank, reduce_scatter, all_gather
        # what is left of the 1/world average after the exchange: the bf16 payload is pre-scaled, the fp32 one (DDP's) is not
        self.post_scale = 1.0 if payload == 'bf16' else 1.0 / world
        self.scratch = {}          # id(flat) -> bf16 bucket buffer of the same length
        self.stage = {}            # (id(flat), lo) -> (staging buffer [world * c], reduced chunk [c])  (rs_ag)
        self.owned = {}            # id(flat) -> {lo: (a, b)}: the piece of range [lo, hi) this rank owns after the reduce-scatter
        self.ranges = {}           # id(flat) -> {lo: hi}
        self.log = []              # [(tag, lo, hi, n_collectives)] in issue order (tests assert the overlap structure on this)
        self.bytes_per_step = 0
        self.timing = False        # True: every issue() is bracketed by events on the communication stream (bench.py --gpus N)
        self.events = []           # [(start, end)] of the current step
        self.comm_ms = []          # per finished step: summed duration of its issue() brackets (busy time of the communication stream)
        # timing events: objects with record() / elapsed_time(other) in ms (device events; tests inject a fake clock)
        self.event_factory = event_factory or (lambda: torch.cuda.Event(enable_timing=True))

    def describe(self):
        return dict(payload=self.payload, transport=self.transport, mode=self.mode, chunk_mb=self.chunk * (2 if self.payload == 'bf16' else 4) >> 20,
                    collectives_per_step=sum(e[3] for e in self.log_last), bytes_per_step=self.bytes_last,
                    ranges_per_step=len(self.log_last), post_scale=self.post_scale)

    log_last, bytes_last = (), 0

    def begin_step(self):
        if self.log:
            self.log_last, self.bytes_last = tuple(self.log), self.bytes_per_step
        self.log, self.bytes_per_step = [], 0
        if self.timing and self.events:
            pend = self.events
            self.events = []
            self._pending = getattr(self, '_pending', []) + [pend]

    def collect_timing(self):
        """(after a device synchronise) ms the communication stream was busy, per recorded step"""
        steps = getattr(self, '_pending', []) + ([self.events] if self.events else [])
        self._pending, self.events = [], []
        self.comm_ms += [sum(a.elapsed_time(b) for a, b in st) for st in steps]
        return self.comm_ms

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
            if self.timing:
                ev0, ev1 = self.event_factory(), self.event_factory()
                ev0.record()
            if self.mode == 'rs_ag':
                self._issue_rs(flat, lo, hi)
                n = 1
            else:
                for a, b in bucket_ranges(hi, self.chunk, lo):
                    if self.payload == 'bf16':
                        buf = self._bucket(flat)
                        self.pack(flat[a:b], buf[a:b], 1.0 / self.world)
                        self.all_reduce(buf[a:b])
                        self.unpack(buf[a:b], flat[a:b])
                        self.bytes_per_step += 2 * (b - a)
                    else:
                        self.all_reduce(flat[a:b])
                        self.bytes_per_step += 4 * (b - a)
                    n += 1
            if self.timing:
                ev1.record()
                self.events.append((ev0, ev1))
        self.log.append((tag, lo, hi, n))

    # ---------------------------------------------------------------------------------------- reduce-scatter / all-gather
    def piece(self, lo, hi):
        """[a, b): the part of range [lo, hi) rank `self.rank` owns (chunk r of `world` equal chunks; the last ones may be short
        or empty)"""
        c = rs_chunk(hi - lo, self.world)
        a = min(lo + self.rank * c, hi)
        return a, min(a + c, hi)

    def _issue_rs(self, flat, lo, hi):
        n, W = hi - lo, self.world
        c = rs_chunk(n, W)
        key = (id(flat), lo)
        if key not in self.stage:
            dt = torch.bfloat16 if self.payload == 'bf16' else torch.float32
            self.stage[key] = (torch.zeros(W * c, dtype=dt, device=flat.device), torch.empty(c, dtype=dt, device=flat.device))
        stage, red = self.stage[key]
        if self.payload == 'bf16':
            self.pack(flat[lo:hi], stage[:n], 1.0 / W)                  # (the pad tail of the staging buffer stays zero)
        else:
            _copy_flat(stage[:n], flat[lo:hi])
        self.reduce_scatter(red, stage)
        a, b = self.piece(lo, hi)
        if b > a:
            if self.payload == 'bf16':
                self.unpack(red[:b - a], flat[a:b])
            else:
                _copy_flat(flat[a:b], red[:b - a])
        self.owned.setdefault(id(flat), {})[lo] = (a, b)
        self.ranges.setdefault(id(flat), {})[lo] = hi
        self.bytes_per_step += stage.element_size() * n * (W - 1) // W

    def gather(self, flat_grad, values):
        """all-gather of `values` (fp32 master parameters, same indexing as the gradient buffer `flat_grad` whose ranges were
        reduce-scattered): every rank contributes the pieces it owns and receives everybody else's.  Enqueued on the CURRENT stream."""
        W = self.world
        for lo, hi in sorted(self.ranges.get(id(flat_grad), {}).items()):
            n = hi - lo
            c = rs_chunk(n, W)
            key = ('ag', id(flat_grad), lo)
            if key not in self.stage:
                self.stage[key] = (torch.empty(W * c, dtype=values.dtype, device=values.device), torch.zeros(c, dtype=values.dtype, device=values.device))
            full, mine = self.stage[key]
            a, b = self.piece(lo, hi)
            if b > a:
                _copy_flat(mine[:b - a], values[a:b])
            self.all_gather(full, mine)
            _copy_flat(values[lo:hi], full[:n])
            self.bytes_per_step += values.element_size() * n * (W - 1) // W


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def predict_exchange(segments_ms, stage_bytes, link_gb_s, world, tail_ms=0.0, latency_us=20.0, mode='allreduce'):
    """First-order timeline of the staged exchange on one node (what the first SCALE run is judged against).

    segments_ms : compute time of the backward segments in issue order (the gradients of stage i are complete when segment i ends)
    stage_bytes : payload bytes handed to the communication stream after segment i
    link_gb_s   : per-link bandwidth a ring collective sustains (xGMI: every GPU sends on ONE link per ring step; a ring all-reduce moves
                  2 (W-1)/W of the payload over it, a reduce-scatter or all-gather (W-1)/W)
    The communication stream is serial and starts a stage's collectives when (a) the segment that produced it has ended and (b) the
    previous stage's collectives are done.  Returns comm_ms_total (busy time), comm_ms_exposed (what the compute stream waits for at
    the join behind the last segment) and step_ms = compute + exposed + tail."""
    assert len(segments_ms) == len(stage_bytes) and world >= 1
    factor = 0.0 if world == 1 else (2.0 if mode == 'allreduce' else 1.0) * (world - 1) / world
    t_compute, t_comm, busy = 0.0, 0.0, 0.0
    for seg, nbytes in zip(segments_ms, stage_bytes):
        t_compute += seg
        if nbytes <= 0 or world == 1:
            continue
        dur = nbytes * factor / (link_gb_s * 1e9) * 1e3 + latency_us * 1e-3
        t_comm = max(t_comm, t_compute) + dur
        busy += dur
    exposed = max(0.0, t_comm - t_compute)
    return dict(comm_ms_total=busy, comm_ms_exposed=exposed, step_ms=t_compute + exposed + tail_ms)
