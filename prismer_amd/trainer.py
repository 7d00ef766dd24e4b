"""Native training step for Prismer caption / VQA fine-tuning on MI355X.

Replaces the per-iteration body of the reference training scripts (train_caption.py:126-135, train_vqa.py:118-134):
    cosine_lr_schedule -> loss = model(...) -> optimizer.zero_grad() -> accelerator.backward(loss) -> optimizer.step()
with a hand-scheduled step that never enters autograd:

    [memset grads] -> encoder program fwd -> decoder program fwd (+CE) -> decoder program bwd
        -> (RCCL all-reduce of the decoder's flat fp32 gradient buffer, on the communication stream,
            overlapping the whole encoder backward)
    -> encoder program bwd -> (all-reduce of the encoder gradients) -> fused AdamW over the two flat buffers.

The compute segments are captured once into hipGraphs (torch.cuda.CUDAGraph) and replayed, so the ~1100 kernel launches of
a step cost a handful of graph launches on the host; the collectives stay outside the graphs (eager RCCL calls ordered by
stream events), which keeps the multi-GPU path identical to the single-GPU one plus three all_reduce sweeps.  With more
than one rank the encoder backward is cut after the trunk (Trainer._schedule): the trunk's gradients are exchanged while the
stems' backward runs, only the stems' own gradients wait on the critical path.
Data parallel semantics = DDP's: every rank holds all parameters, gradients are summed over ranks in fp32 and divided by
world size (folded into the AdamW kernel as grad_scale; `grad_payload='bf16'` is an opt-in that halves the bytes: each rank
pre-scales by 1/world and rounds to bf16 before the sum), BatchNorm uses per-rank batch statistics (no SyncBN in the
reference).  DDP's per-step buffer broadcast from rank 0 is replaced by rank-local BN running statistics by default
(documented deviation); Trainer(broadcast_buffers=True) restores DDP's semantics with two collectives per step.
"""
import contextlib
import math
import random
import time

import torch

from . import ops
from .dist import GradExchange, contiguous_stages
from .store import ALIGN

F32, BF16 = torch.float32, torch.bfloat16


from .schedules import cosine_lr                          # noqa: E402  (utils.py:13-17 of the reference; re-exported)


class Trainer:
    def __init__(self, model, lr=5e-5, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, min_lr=0.0, total_steps=1000,
                 task='caption', use_graph=True, process_group=None, bucket_mb=64, side_stream=False, micro_batches=1, keep_grads=False,
                 max_text_len=None, grad_payload='fp32', transport='torch.distributed', dec_backward_stages=3, lr_schedule=None,
                 shard_optimizer=False, overwrite_single_writer=True, allow_eager_fallback=False, broadcast_buffers=False, exchange_issue='device'):
        self.model = model
        self.enc, self.dec = model.expert_encoder, model.text_decoder
        self.init_lr, self.min_lr, self.total_steps = lr, min_lr, total_steps
        # lr(it) for it = 0, 1, ...: default = the fine-tuning scripts' per-iteration cosine; prismer_amd.schedules.pretrain_schedule
        # gives train_pretrain.py's epoch-cosine + linear warm-up
        self.lr_schedule = lr_schedule or (lambda it: cosine_lr(it, self.total_steps, self.init_lr, self.min_lr))
        self.wd, self.betas, self.eps = weight_decay, betas, eps
        self.task = task
        self.max_text_len = max_text_len
        self.use_graph = use_graph
        self.pg = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        self.it = 0
        self.graphs = None
        # How a finished stage's gradients reach the communication stream (N > 1, graph replay):
        #   'device' (default): the communication stream waits for an event of the compute stream (round 1-6 behaviour);
        #   'host': the NEXT segment is enqueued first, then the host waits for the stage's event and launches the collectives with no
        #           device-side edge.  tools/stream_edge_probe.py (one GPU, stand-in kernels): an edge from the graph-launching stream to another
        #           stream costs the replayed step a fixed ~0.45 ms, a host wait with the next segment already enqueued nothing.  Not the default
        #           because no RCCL run has compared the two yet; identical arithmetic, identical collective order.
        if exchange_issue not in ('device', 'host'):
            raise ValueError("exchange_issue must be 'device' or 'host'")
        self.exchange_issue = exchange_issue
        self._host_wait_s = 0.0
        self.static = None
        self.staging = None                    # second input buffer set of prefetch_batch / commit_prefetched
        self._staged = False                   # a batch sits in the staging set, not yet committed
        self.enc.train(); self.dec.train()
        self.enc_prog, self.dec_prog = self.enc._program(), self.dec._program()
        self.stores = [self.enc._store, self.dec._store]
        for st in self.stores:
            st.native_grads = True
            st.managed = True
            st._grad_cur = st.grad
        dev = self.stores[0].master.device
        self.device = dev
        # shard_optimizer (ZeRO-1 style, the role of accelerate's --shard_grad_op flag in train_pretrain.py:56-91,104-107): every
        # rank keeps Adam moments for, and updates, only ITS contiguous 1/world slice of each flat buffer, then the owners
        # broadcast their updated fp32 slices.  Optimizer state and AdamW time shrink by 1/world; off by default.
        # shard_optimizer='rs_ag' (round 3; the communication pattern of FSDP SHARD_GRAD_OP, train_caption.py:56-66): gradients are
        # REDUCE-SCATTERED per finished range (each rank receives 1/world of every range), AdamW runs on the owned pieces only and the
        # updated fp32 parameters are ALL-GATHERED -- half the bytes of an all-reduce per link for the gradients, optimizer state and
        # update 1/world, no broadcast fan-out.
        self.rank = torch.distributed.get_rank(process_group) if self.world > 1 else 0
        self.rs_ag = shard_optimizer == 'rs_ag' and self.world > 1
        self.shard = bool(shard_optimizer) and not self.rs_ag and self.world > 1
        own = [self._shard_bounds(i)[self.rank] for i in range(len(self.stores))] if self.shard else [(0, st.n_train) for st in self.stores]
        if self.rs_ag:
            own = [(0, 0) for _ in self.stores]                # moments live per owned piece (self.piece_state), allocated on first use
        self.m = [torch.zeros(max(hi - lo, 1), dtype=F32, device=dev) for lo, hi in own]
        self.v = [torch.zeros(max(hi - lo, 1), dtype=F32, device=dev) for lo, hi in own]
        self.piece_state = {}                                  # rs_ag: (store index, a, b) -> (m, v)
        self.hyper = torch.zeros(3, dtype=F32, device=dev)
        # per-step host values (learning rate, Adam bias corrections, instance draws) travel in the arguments of one tiny kernel
        # (ops.store_words): stream-ordered like the copies they replace, no pinned staging, no copy engine
        self.table = torch.zeros(256, dtype=torch.int32, device=dev)
        self.seed = self.dec.dropout_seed()
        self.comm_stream = torch.cuda.Stream(device=dev) if self.world > 1 else None
        # optimizer stream: with N ranks the decoder's AdamW runs beside the encoder backward as soon as its buckets are reduced (an
        # eager launch between graph replays).  One rank: measured neutral as a graph branch (28.99 vs 29.03 ms) and +0.4 ms as an eager
        # launch on a second stream, so it simply runs after the encoder backward.
        self.opt_stream = torch.cuda.Stream(device=dev) if self.world > 1 else None
        self.micro = micro_batches if (side_stream and micro_batches > 1) else 1
        nl = len(self.dec_prog.layers)
        k = max(1, min(dec_backward_stages, nl)) if (self.world > 1 and self.micro == 1) else 1
        self.dec_cuts = [nl - (nl * i) // k for i in range(k + 1)]      # e.g. 12 layers, 3 stages: [12, 8, 4, 0]
        self._bucket_mb = bucket_mb
        self.bucket_elems = bucket_mb * 1024 * 1024 // (2 if grad_payload == 'bf16' else 4)
        self.exchange = self._make_exchange(grad_payload, transport)
        self.stage_ranges = self._stage_ranges()
        # Micro-batches (optional, default off): the stems see the whole batch (train-mode BatchNorm statistics), everything
        # after them is sample-independent, so the batch can be cut into `micro_batches` contiguous slices that run as
        # parallel branches (one stream each).  Motivation: at bs32 the phase times shrink only ~0.7x when the batch is
        # halved, i.e. most kernels under-fill the 256 CUs.  Measured on MI355X / ROCm 7.2 (same box, bs32): hipGraph replay
        # 33.2 ms with 1 slice vs 44.7 ms with 2 (this runtime's graph executor handles wide parallel branches poorly;
        # eager launches with 2 slices: 38.5 ms, host-bound) -- so the default stays 1.  Weight-gradient accumulation is
        # race-free either way because every read-modify-write of the gradient buffer lives on the single side stream, and
        # forked streams never re-join work they forked themselves (capture_end crashes on such diamonds here).
        # side_stream=True forks independent work (deferred weight gradients, the six stems, the cross-attention K/V projections)
        # onto extra streams = parallel branches of the captured graphs.  Default OFF since round 2: it bought 0.8 % of the step,
        # and under hipGraph REPLAY on ROCm 7.2 the graphs with forked branches mis-ordered the deferred LayerNorm gamma/beta
        # reduction against the LayerNorm backward kernels that feed it (decoder LayerNorm parameter gradients came out as
        # uncorrelated garbage -- caught by tests/test_parity_gpu.py::test_trainer_hipgraph_step_matches_reference_golden; the same
        # kernels launched eagerly on the same streams, and graphs without branches, are exact).  A branch-free graph is a chain:
        # every kernel depends on its predecessor, nothing is left to the executor.
        if side_stream and use_graph:
            raise RuntimeError('Trainer(side_stream=True, use_graph=True): graphs with forked branches are not replayed correctly on this '
                               'ROCm (see the comment above); use one of the two')
        if side_stream:
            ops.SIDE = ops.SideStream(dev)
            ops.POOL = ops.BranchPool(dev, 3)
            ops.MICRO = ops.BranchPool(dev, self.micro) if self.micro > 1 else ops._NoPool()
        else:
            ops.SIDE, ops.POOL, ops.MICRO = None, ops._NoPool(), ops._NoPool()
        self.loss = None
        self.exposed_events = []
        self.trace = []
        self._grads_clean = False
        self.keep_grads = keep_grads           # True: gradients stay readable after step() (tests); False: AdamW zeroes them
        self._overwrite = bool(overwrite_single_writer)          # see _wq_scope
        self.allow_eager_fallback = bool(allow_eager_fallback)
        self._exclusive, self._keep_maps, self._count_config = None, None, None
        self._graph_no_fill = False
        self._host_comm_s, self._host_comm_steps = 0.0, 0
        # DDP(broadcast_buffers=True) semantics (what accelerate wraps the model in, train_caption.py:117): before every forward rank 0's BatchNorm
        # running statistics and counters overwrite everybody's.  Off by default: rank-local statistics, the documented deviation of SURVEY 8e
        # (a per-step collective in front of the forward for buffers no training computation reads)
        self.broadcast_buffers = bool(broadcast_buffers) and self.world > 1
        self._step_open = False                # a step was started and did not reach its last segment (exception between replays)
        if self.world > 1:
            self.broadcast_parameters()

    # ------------------------------------------------------------------------------------------ distributed
    def broadcast_parameters(self):
        """DDP construction semantics (train_caption.py:117): rank 0's parameters and buffers win."""
        for st in self.stores:
            torch.distributed.broadcast(st.master, 0, group=self.pg)
            st.refresh()
        for mod in (self.enc, self.dec):
            for b in mod.buffers():
                torch.distributed.broadcast(b, 0, group=self.pg)

    def sync_buffers(self):
        """rank 0's BatchNorm running statistics and num_batches_tracked -> every rank (DDP's _sync_module_buffers before each forward), two
        collectives on the compute stream: the programs keep these buffers as views of two flat tensors"""
        ep = self.enc_prog
        if self.world == 1 or ep._bn_stats_flat is None:
            return
        self._bcast(ep._bn_stats_flat, 0)
        self._bcast(ep._bn_flat.view(torch.float32) if self._native_comm is not None else ep._bn_flat, 0)

    def _make_exchange(self, payload, transport):
        if self.world == 1:
            return None
        self._native_comm = None
        if transport == 'native':                              # the library's own RCCL communicator (include/prismer_comm.h)
            from . import comm
            self._native_comm = comm.NativeComm.from_process_group(self.pg, self.device)
            nc = self._native_comm
            reduce_fn, rs_fn, ag_fn = nc.all_reduce_, nc.reduce_scatter, nc.all_gather
            # (round 6: reduce-scatter / all-gather / broadcast are entry points of the communicator too, so the sharded modes run on ONE
            # transport -- a second RCCL communicator driven from another stream is the classic multi-communicator ordering hazard)
            self._bcast = lambda t, src: nc.broadcast_(t, src)
        else:
            def reduce_fn(t):
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.pg)

            def rs_fn(out, inp):
                torch.distributed.reduce_scatter_tensor(out, inp, op=torch.distributed.ReduceOp.SUM, group=self.pg)

            def ag_fn(out, inp):
                torch.distributed.all_gather_into_tensor(out, inp, group=self.pg)

            def _bcast(t, src):
                torch.distributed.broadcast(t, torch.distributed.get_global_rank(self.pg, src) if self.pg is not None else src, group=self.pg)
            self._bcast = _bcast
        if payload == 'auto':
            payload = self._choose_payload(reduce_fn)
            self.bucket_elems = self._bucket_mb * 1024 * 1024 // (2 if payload == 'bf16' else 4)
        return GradExchange(self.world, reduce_fn, pack=lambda src, dst, scale=1.0: ops.cast_to_bf16(src, out=dst, scale=scale),
                            unpack=lambda src, dst: ops.cast_to_f32(src, out=dst), payload=payload, chunk_elems=self.bucket_elems,
                            comm_stream=self.comm_stream, transport='rccl via ' + ('prismer_comm (native)' if transport == 'native' else 'torch.distributed'),
                            mode='rs_ag' if self.rs_ag else 'allreduce', rank=self.rank, reduce_scatter=rs_fn, all_gather=ag_fn)

    AUTO_BUSBW_GB_S = 150.0          # below this measured all-reduce bus bandwidth the fp32 payload no longer hides behind the backward

    def _choose_payload(self, reduce_fn):
        """grad_payload='auto' (bench.py default for N > 1, round 6): DDP's fp32 payload when the node's all-reduce sustains it, else the
        pre-scaled bf16 buckets (half the bytes per link).  Decided ONCE at construction from a probe all-reduce -- 3 x 64 MB fp32 on the
        communication stream, bus bandwidth = 2 (W-1)/W x bytes / time, the slowest rank's time (MAX all-reduce) so that every rank decides alike --
        and shown beside dist.predict_exchange on the measured phase times of one GPU; small worlds (< 4 ranks) keep fp32.  The decision and the probe
        are part of exchange_desc() / the bench line."""
        from .dist import predict_exchange
        self.payload_decision = dict(requested='auto', world=self.world)
        if self.world < 4 or self.device.type != 'cuda':
            self.payload_decision.update(chosen='fp32', reason='world < 4: DDP payload')
            return 'fp32'
        n = 16 << 20
        buf = torch.zeros(n, dtype=F32, device=self.device)
        with torch.cuda.stream(self.comm_stream):
            reduce_fn(buf)                                         # channel set-up
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                reduce_fn(buf)
            b.record()
        b.synchronize()
        t = torch.tensor([a.elapsed_time(b) / 3.0], dtype=torch.float64, device=self.device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.pg)
        ms = float(t.item())
        busbw = 2.0 * (self.world - 1) / self.world * n * 4 / (ms * 1e-3) / 1e9
        # phase times of one MI355X at Prismer-BASE bs32 (profiles/r5_phase_times.txt): decoder backward stages, trunk, front; payload per stage (fp32 MB)
        seg, mb = [1.6, 1.1, 1.3, 7.0, 3.2], [182, 151, 364, 170, 102]
        link = busbw * self.world / (2.0 * (self.world - 1))
        pred = {p: predict_exchange(seg, [m * 1e6 * (0.5 if p == 'bf16' else 1.0) for m in mb], link, self.world) for p in ('fp32', 'bf16')}
        chosen = 'fp32' if busbw >= self.AUTO_BUSBW_GB_S else 'bf16'
        self.payload_decision.update(chosen=chosen, probe_ms_64mb=round(ms, 3), probe_busbw_gb_s=round(busbw, 1), threshold_gb_s=self.AUTO_BUSBW_GB_S,
                                     predicted_exposed_ms={p: round(v['comm_ms_exposed'], 2) for p, v in pred.items()})
        return chosen

    def exchange_desc(self):
        if self.exchange is None:
            return None
        d = self.exchange.describe()
        d['issue'] = self.exchange_issue + (' (stream event edge per stage)' if self.exchange_issue == 'device' else ' (host waits for the stage, next segment already enqueued)')
        if getattr(self, 'payload_decision', None):
            d['payload_decision'] = self.payload_decision
        return d

    def _stage_ranges(self):
        """{stage: [(store index, lo, hi)]}: the ranges of the flat gradient buffers whose gradients are COMPLETE once backward
        stage `stage` has run, in buffer order.  Decoder stages follow the reverse layer order of its backward -- 'dec0' = LM
        head, output_layer and the last layers ... 'dec{k-1}' = the first layers, the embeddings (tied LM-head weight: written
        at both ends of the backward) and the merged cross-attention K/V projection (one wgrad after the last layer);
        'trunk' = ViT adaptors, resampler, ln_pre / ln_post; 'front' = expert stems, positional / instance embeddings."""
        enc, dec = self.stores
        cuts = self.dec_cuts
        merged_kv = self.dec_prog.kv_all is not None
        last = f'dec{len(cuts) - 2}'

        def dec_stage(n):
            if n.startswith('lm_head.') or n.startswith('roberta.encoder.output_layer.'):
                return 'dec0'
            if n.startswith('roberta.encoder.layer.'):
                l = int(n.split('.')[3])
                if merged_kv and ('.1.self.key.' in n or '.1.self.value.' in n):
                    return last
                for k in range(len(cuts) - 1):
                    if cuts[k + 1] <= l < cuts[k]:
                        return f'dec{k}'
            return last                                            # embeddings

        def enc_stage(n):
            return 'front' if n.startswith(('positional_embedding', 'instance_embedding', 'conv1.')) else 'trunk'
        out = {}
        for si, (st, fn) in enumerate(((enc, enc_stage), (dec, dec_stage))):
            tr = [n for n in st.names if st.is_trainable(n)]
            for stage, lo, hi in contiguous_stages(tr, st.offset, st.numel, fn, ALIGN):
                out.setdefault(stage, []).append((si, lo, min(hi, st.n_train)))
        return out

    def _issue(self, stage):
        """hand the finished ranges of `stage` to the gradient exchange (communication stream, behind an event recorded now)"""
        if self.exchange is None:
            return
        self.trace.append(('issue', stage))
        ev, self._issue_event = self._issue_event, None       # host mode: recorded right behind the stage's replay, before the next segment was enqueued
        if ev is not None:
            t0 = time.perf_counter()
            ev.synchronize()                                   # the stage's gradients are complete: no device-side edge needed
            self._host_wait_s += time.perf_counter() - t0     # (waiting for COMPUTE: not communication time, kept out of comm_ms_exposed)
            after = None
        else:                                                  # device mode, eager steps, the warm-up passes of the capture
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream())
        for si, lo, hi in self.stage_ranges.get(stage, ()):
            self.exchange.issue(self.stores[si].grad, lo, hi, tag=f'{stage}:{si}', after=after)

    _issue_event = None

    def _wait_comm(self):
        if self.world > 1:
            cur = torch.cuda.current_stream()
            if self.exchange is not None and self.exchange.timing:
                # exposed communication = how long the compute stream sits in this join (everything before it overlapped the backward)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cur)
                cur.wait_stream(self.comm_stream)
                b.record(cur)
                self.exposed_events.append((a, b))
                return
            cur.wait_stream(self.comm_stream)

    exposed_events = []

    def comm_timing(self):
        """(bench.py --gpus N) after a device synchronise: mean ms per step the communication stream was busy, and the part of it the
        compute stream had to wait for at the join before the optimizer (the rest overlapped the backward)"""
        if self.exchange is None:
            return None
        busy = self.exchange.collect_timing()
        exposed = [a.elapsed_time(b) for a, b in self.exposed_events]
        self.exposed_events = []
        n = max(len(exposed), 1)
        # exposed = what the compute stream waited at the join (device events) + what the HOST spent inside the exchange calls (round 6: the
        # gloo dry run blocks the host for ~300 ms per step and reported 0.07 ms exposed, because the device never saw the wait)
        host = self._host_comm_s * 1e3 / max(self._host_comm_steps, 1)
        self._host_comm_s, self._host_comm_steps = 0.0, 0
        join = sum(exposed) / n
        return dict(comm_ms_total=round(sum(busy) / max(len(busy), 1), 3), comm_ms_exposed=round(join + host, 3), comm_ms_exposed_device_join=round(join, 3),
                    comm_ms_host_in_exchange_calls=round(host, 3), steps_timed=len(exposed))

    # ------------------------------------------------------------------------------------------ step pieces
    def _slices(self, B):
        n = self.micro if (self.micro > 1 and B >= 2 * self.micro) else 1
        cuts = [B * i // n for i in range(n + 1)]
        return [(cuts[i], cuts[i + 1]) for i in range(n)]

    def _seg_forward(self, s, dec_hi_lo):
        """forward of everything + the first stage of the decoder backward (CE, LM head, output_layer, layers hi-1 .. lo)"""
        # optimizer.zero_grad(): the fused AdamW of the previous step already left the gradient buffers zeroed (they are
        # allocated zeroed); only a step that did not end in the optimizer segments (first step after an exception) needs the fill
        if not self._grads_clean:
            for st in self.stores:
                st.grad.zero_()
        self._grads_clean = False
        ep, dp = self.enc_prog, self.dec_prog
        d = ep.d
        S, Mx = d.seq_len, d.num_expert_tokens
        h, xf, self.sv_f = ep.forward_front(s['experts'], self.table, True, True)
        B = s['input_ids'].shape[0]
        parts = self._slices(B)
        self.sv_t, self.denc, self.dec_state, losses = [None] * len(parts), [None] * len(parts), [None] * len(parts), [None] * len(parts)
        dp.kv_prefetch = len(parts) == 1
        for mi, (b0, b1) in enumerate(parts):
            with (ops.MICRO.branch(mi) if len(parts) > 1 else contextlib.nullcontext()):
                Bh = b1 - b0
                enc_out, self.sv_t[mi] = ep.forward_trunk(h[b0 * S:b1 * S], None if xf is None else xf[b0 * Mx:b1 * Mx], Bh, True)
                dp.site_base = mi << 20
                _, loss, sv_d = dp.forward(s['input_ids'][b0:b1], s['attention_mask'][b0:b1], enc_out, s['labels'][b0:b1], self.seed, True)
                # caption: loss.mean() (prismer_caption.py:33); VQA: (weights * loss).mean() (prismer_vqa.py:40-41).  d(total)/d(loss_b) =
                # weights_b / B is a static input (set_batch); the total is one ph_weighted_sum_f32 launch with scale 1/B
                w = None if s.get('weights') is None else s['weights'][b0:b1]
                dloss, losses[mi] = s['dloss'][b0:b1], ops.weighted_sum(loss, w, 1.0 / B)
                st = dp.backward_start(sv_d, dloss)
                dp.backward_layers(st, *dec_hi_lo)
                if dec_hi_lo[1] == 0:
                    self.denc[mi] = dp.backward_finish(st)     # (valid after the join_side below)
                else:
                    self.dec_state[mi] = st
                del sv_d, st
        dp.site_base = 0
        if len(parts) > 1:
            ops.MICRO.join()
        ops.join_side()
        self.loss_buf = losses[0][0] if len(losses) == 1 else torch.stack(losses).sum()      # (micro-batch slices: eager-only experiment)

    def _seg_dec_backward(self, hi, lo):
        """decoder layers hi-1 .. lo of the backward (+ embeddings / merged K/V when lo == 0); single-slice schedules only"""
        dp = self.dec_prog
        st = self.dec_state[0]
        dp.backward_layers(st, hi, lo)
        if lo == 0:
            self.denc[0] = dp.backward_finish(st)
            self.dec_state = None
        ops.join_side()

    def _seg_enc_trunk_backward(self):
        ep = self.enc_prog
        d = ep.d
        S, Mx = d.seq_len, d.num_expert_tokens
        parts = [(sv['B']) for sv in self.sv_t]
        B = sum(parts)
        dev = self.denc[0].device
        self.dh = torch.empty(B * S, d.width, dtype=BF16, device=dev)
        self.dxf = torch.empty(B * Mx, d.width, dtype=BF16, device=dev) if self.sv_t[0]['has_x'] else None
        b0 = 0
        for mi, Bh in enumerate(parts):
            b1 = b0 + Bh
            with (ops.MICRO.branch(mi) if len(parts) > 1 else contextlib.nullcontext()):
                ep.backward_trunk(self.sv_t[mi], self.denc[mi], self.dh[b0 * S:b1 * S],
                                  None if self.dxf is None else self.dxf[b0 * Mx:b1 * Mx])
            b0 = b1
        if len(parts) > 1:
            ops.MICRO.join()
        self.sv_t = self.denc = None

    def _seg_enc_front_backward(self):
        self.enc_prog.backward_front(self.sv_f, self.dh, self.dxf)
        ops.join_side()
        self.sv_f = self.dh = self.dxf = None

    def _shard_bounds(self, i):
        """[(lo, hi)] per rank: contiguous, 256-element aligned slices of the trainable range of store i"""
        n, W = self.stores[i].n_train, self.world
        per = ((n + W - 1) // W + 255) // 256 * 256
        return [(min(k * per, n), min((k + 1) * per, n)) for k in range(W)]

    def _adamw_sharded(self, i):
        """eager (between graph replays): AdamW on the own slice of the reduced gradients, owners broadcast their fp32 slices, bf16
        shadows re-derived.  Collectives use torch.distributed on the current stream."""
        st = self.stores[i]
        bounds = self._shard_bounds(i)
        lo, hi = bounds[self.rank]
        if hi > lo:
            ops.adamw(st.master[lo:hi], st.grad[lo:hi], self.m[i], self.v[i], st.shadow[lo:hi], hi - lo, self.hyper, self.betas[0],
                      self.betas[1], self.eps, self.wd, self._post_scale(), zero_grad=False)
        for k, (a, b) in enumerate(bounds):
            if b > a:
                self._bcast(st.master[a:b], k)
        ops.cast_to_bf16(st.master[:st.n_train], st.shadow[:st.n_train])
        st.refresh_derived()

    def _adamw(self, i):
        st = self.stores[i]
        ops.adamw(st.master, st.grad, self.m[i], self.v[i], st.shadow, st.n_train, self.hyper, self.betas[0], self.betas[1], self.eps,
                  self.wd, self._post_scale(), zero_grad=not self.keep_grads, keep=self._keep_maps[i] if self._keep_maps else None)
        st.refresh_derived()

    # ------------------------------------------------------------------------------------------ single-writer weight gradients
    @contextlib.contextmanager
    def _wq_scope(self):
        """Brackets one whole step (eager, capture warm-up or capture).  The first bracketed step COUNTS the deferred weight-gradient
        GEMMs per output; outputs inside the flat gradient buffers that are written exactly once per step are from then on
        OVERWRITTEN by their GEMM (accumulate = 0: no read of the zeroed buffer) and kept out of AdamW's zero_grad stores (keep bitmap,
        whole 1024-element chunks only) -- 0.9 GB less read and 0.9 GB less written per step at Prismer-BASE.  Everything else
        (embeddings, biases, LayerNorm / BatchNorm parameters, conv weights, tied or micro-batched weights) keeps accumulate-into-zero.
        Trainer(overwrite_single_writer=False) switches it off."""
        wq = ops.WQ
        if not self._overwrite:
            yield
            return
        if self._exclusive is not None and self._count_config != self._wq_config():
            # the frozen single-writer state belongs to another configuration (queue switched, micro-batching changed): recount
            self._exclusive, self._keep_maps = None, None
        counting = self._exclusive is None
        if counting:
            wq.counts = {}
        else:
            wq.exclusive = self._exclusive
        completed = False
        try:
            yield
            completed = True
        finally:
            # only a step that ran to its end has seen every writer: a partial count (failed capture warm-up, an exception in a
            # collective) would mark multiply-written outputs exclusive and later steps would overwrite instead of accumulate
            if counting and completed and wq.counts is not None:
                self._finish_count(wq.counts)
                self._count_config = self._wq_config()
            wq.counts, wq.exclusive = None, None

    def _wq_config(self):
        """what the frozen single-writer state depends on: a change invalidates it (see _wq_scope)"""
        return (bool(ops.WQ.enabled), self.micro, self.keep_grads, tuple(st.n_train for st in self.stores))

    def _finish_count(self, counts):
        excl = set()
        maps = []
        self._excl_ranges = []                                 # per store: [(offset, numel)] of the overwritten outputs (tests, diagnostics)
        for st in self.stores:
            self._excl_ranges.append([])
            base, n = st.grad.data_ptr(), st.n_train
            nchunks = (n + 1023) // 1024
            keep = torch.zeros((nchunks + 31) // 32, dtype=torch.int32)
            bits = keep.numpy().view('uint32')
            inside = sorted(((key - base) // 4, numel, key, cnt, ok) for key, (cnt, numel, ok) in counts.items()
                            if base <= key < base + 4 * n and (key - base) % 4 == 0)
            for k, (off, numel, key, cnt, ok) in enumerate(inside):
                alone = (k == 0 or inside[k - 1][0] + inside[k - 1][1] <= off) and (k + 1 == len(inside) or off + numel <= inside[k + 1][0])
                if cnt != 1 or not ok or off + numel > n or not alone:             # (overlapping outputs: two writers of the shared part)
                    continue
                excl.add(key)
                self._excl_ranges[-1].append((off, numel))
                for c in range((off + 1023) // 1024, (off + numel) // 1024):       # chunks entirely inside the output
                    bits[c >> 5] |= (1 << (c & 31))
            maps.append(keep.to(self.device))
        self._exclusive, self._keep_maps = excl, maps

    def _post_scale(self):
        """what is left of DDP's 1/world average after the exchange (1/world for the fp32 SUM, 1 for the pre-scaled bf16 payload)"""
        return 1.0 if self.exchange is None else self.exchange.post_scale

    def _adamw_rs(self, i):
        """rs_ag: AdamW on the pieces of store i this rank received from the reduce-scatters, all-gather of the updated fp32
        parameters, bf16 shadows re-derived.  Eager, on the current stream (collectives cannot sit inside a captured segment)."""
        st = self.stores[i]
        ex = self.exchange
        for lo, (a, b) in sorted(ex.owned.get(id(st.grad), {}).items()):
            if b <= a:
                continue
            key = (i, a, b)
            if key not in self.piece_state:
                self.piece_state[key] = (torch.zeros(b - a, dtype=F32, device=self.device), torch.zeros(b - a, dtype=F32, device=self.device))
            m, v = self.piece_state[key]
            ops.adamw(st.master[a:b], st.grad[a:b], m, v, st.shadow[a:b], b - a, self.hyper, self.betas[0], self.betas[1], self.eps,
                      self.wd, ex.post_scale, zero_grad=False)
        ex.gather(st.grad, st.master)
        ops.cast_to_bf16(st.master[:st.n_train], st.shadow[:st.n_train])
        st.refresh_derived()

    def _tail_rs(self):
        """rs_ag: both updates run after the last reduce-scatter (gradients are re-zeroed by the next step's first segment)"""
        self._adamw_rs(1)
        self._adamw_rs(0)
        ops.advance_seed(self.seed)
        self._grads_clean = False

    def _seg_enc_backward_with_dec_adamw(self):
        """one rank: the decoder's gradients are final when the encoder backward starts, and nothing in the encoder backward
        reads a decoder weight -- its fused AdamW (HBM-bound: 5.1 GB of state for Prismer-BASE) runs as a parallel branch
        beside the MFMA-bound encoder backward instead of after it."""
        main = torch.cuda.current_stream()
        if self.opt_stream is not None:
            ev = torch.cuda.Event()
            ev.record(main)
            self.opt_stream.wait_event(ev)
            with torch.cuda.stream(self.opt_stream):
                self._adamw(1)
        self._seg_enc_trunk_backward()
        self._seg_enc_front_backward()
        if self.opt_stream is not None:
            main.wait_stream(self.opt_stream)
        else:
            self._adamw(1)

    def _seg_enc_trunk_backward_joined(self):
        self._seg_enc_trunk_backward()
        ops.join_side()                                        # trunk gradients complete: their all-reduce may start

    def _seg_optimizer_tail(self):
        """encoder AdamW (the decoder's already ran beside the encoder backward), dropout seed advance"""
        self._adamw(0)
        ops.advance_seed(self.seed)
        self._grads_clean = not self.keep_grads

    def _dec_adamw_after_comm(self):
        """N ranks: the decoder's AdamW starts on the optimizer stream as soon as its last bucket has been reduced and runs
        beside the encoder backward (an eager launch between graph replays, ordered by stream events like the collectives)"""
        self.opt_stream.wait_stream(self.comm_stream)
        with torch.cuda.stream(self.opt_stream):
            if self.shard:
                self._adamw_sharded(1)
            else:
                self._adamw(1)

    def _tail_sharded(self):
        """sharded optimizer: the encoder's update and the seed advance run eagerly after the last bucket (collectives cannot sit
        inside a captured segment); gradients are re-zeroed by the next step's first segment"""
        torch.cuda.current_stream().wait_stream(self.opt_stream)
        self._adamw_sharded(0)
        ops.advance_seed(self.seed)
        self._grads_clean = False

    def _schedule(self):
        """[(compute segment, host action right after it)].
        One rank: the whole step is ONE segment (forward, decoder backward, encoder backward, both AdamW launches, seed advance).
        Data parallel: the backward is cut in reverse layer order -- decoder in `len(dec_cuts) - 1` stages, encoder trunk,
        encoder front; after every segment the finished ranges go to the bf16 all-reduce on the communication stream, so only
        the last one (the stems' 25 M parameters) is exchanged on the critical path; the decoder's AdamW runs on the
        optimizer stream as soon as its buckets are reduced."""
        s = self.static
        if self.world == 1:
            nl = len(self.dec_prog.layers)

            def whole_step():                  # ONE captured graph (round 6: three replays per step cost 0.18 ms in launch gaps, tools/stream_edge_probe.py's box)
                self._seg_forward(self.static, (nl, 0))
                self._seg_enc_backward_with_dec_adamw()
                self._seg_optimizer_tail()
            return [(whole_step, None)]
        cuts = self.dec_cuts
        nst = len(cuts) - 1
        sched = [(lambda: self._seg_forward(self.static, (cuts[0], cuts[1])), lambda: self._issue('dec0'))]
        for k in range(1, nst):
            sched.append(((lambda k=k: self._seg_dec_backward(cuts[k], cuts[k + 1])), (lambda k=k: self._issue(f'dec{k}'))))
        seg, host = sched[-1]
        if not self.rs_ag:
            sched[-1] = (seg, lambda host=host: (host(), self._dec_adamw_after_comm()))
        sched += [(self._seg_enc_trunk_backward_joined, lambda: self._issue('trunk'))]
        if self.rs_ag:
            sched += [(self._seg_enc_front_backward, lambda: (self._issue('front'), self._wait_comm(), self._tail_rs()))]
        elif self.shard:
            sched += [(self._seg_enc_front_backward, lambda: (self._issue('front'), self._wait_comm(), self._tail_sharded()))]
        else:
            # (stream joins are host actions BETWEEN replays: a wait recorded inside a captured segment would be frozen at capture)
            sched += [(self._seg_enc_front_backward, lambda: (self._issue('front'), self._wait_comm(),
                                                              torch.cuda.current_stream().wait_stream(self.opt_stream))),
                      (self._seg_optimizer_tail, None)]
        return sched

    def _host_prologue(self):
        """per-step host work: LR schedule (cosine per ITERATION, train_caption.py:127), Adam bias corrections and the
        instance-embedding draw table (vit.py:145-147: Python `random`), shipped in the arguments of one kernel launch (round 6: the two pinned
        copies they used to ride queued behind a loader's host-to-device prefetch on the copy engine)."""
        self.it += 1
        lr = self.lr_schedule(self.it - 1)
        hyper = (lr, 1.0 - self.betas[0] ** self.it, 1.0 - self.betas[1] ** self.it)
        if 'obj_detection' in self.enc.experts:
            ops.store_words(self.hyper, hyper, self.table, [random.randint(0, 127) for _ in range(256)])
        else:
            ops.store_words(self.hyper, hyper)

    # ------------------------------------------------------------------------------------------ public API
    def set_batch(self, experts, input_ids, attention_mask, labels, weights=None):
        """(re)binds the static input buffers. First call allocates them; later calls copy into them."""
        for k, v in experts.items():                 # expert-map resolution fixes the program's token geometry
            if k != 'rgb':
                er = (v.get('label_map', v.get('label', v.get('raw'))) if isinstance(v, dict) else v).shape[-1]
                if er != self.enc.expert_resolution:
                    assert self.graphs is None, 'expert resolution changed after graph capture'
                    self.enc.expert_resolution, self.enc._prog = er, None
                    self.enc_prog = self.enc._program()
                break
        # The step is a fixed-shape program (hipGraph replay over static buffers).  The reference tokenises with
        # padding='longest' (prismer_caption.py:20, prismer_vqa.py:25-30), so T varies per batch: text is padded here to the
        # static length with <pad> / mask 0 / label -100 -- masked keys and ignored labels, i.e. the same loss.  The batch
        # size is fixed like the reference's training loader (dataset/__init__.py:42 drop_last=True).
        pad = self.dec.config.pad_token_id
        if self.static is None:
            T = max(self.max_text_len or 0, input_ids.shape[1])
            B = input_ids.shape[0]

            def clone(t):
                return {k: clone(v) for k, v in t.items()} if isinstance(t, dict) else t.to(self.device).contiguous().clone()
            self.static = dict(experts=clone(experts),
                               input_ids=torch.full((B, T), pad, dtype=input_ids.dtype, device=self.device),
                               attention_mask=torch.zeros((B, T), dtype=attention_mask.dtype, device=self.device),
                               labels=torch.full((B, T), -100, dtype=labels.dtype, device=self.device),
                               weights=None if weights is None else torch.zeros(B, dtype=F32, device=self.device),
                               dloss=torch.full((B,), 1.0 / B, dtype=F32, device=self.device))      # d(total) / d(loss_b): 1/B or weights_b / B

        self._bind(self.static, experts, input_ids, attention_mask, labels, weights)

    def _bind(self, s, experts, input_ids, attention_mask, labels, weights):
        """copies one batch into the buffer set `s` (the static inputs of the captured step, or the staging set of prefetch_batch)"""
        pad = self.dec.config.pad_token_id

        def copy(dst, src, what):
            if isinstance(dst, dict):
                if not isinstance(src, dict) or set(src) != set(dst):
                    raise ValueError(f'set_batch: {what} keys changed: {sorted(src) if isinstance(src, dict) else type(src)} vs {sorted(dst)}')
                for k in dst:
                    copy(dst[k], src[k], f'{what}[{k!r}]')
                return
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f'set_batch: {what} has shape {tuple(src.shape)}, the bound program expects {tuple(dst.shape)} '
                                 '(fixed batch size / expert resolution; use drop_last=True like the reference loader)')
            dst.copy_(src, non_blocking=True)

        def copy_text(dst, src, fill, what):
            B, T = dst.shape
            if src.shape[0] != B or src.shape[1] > T:
                raise ValueError(f'set_batch: {what} has shape {tuple(src.shape)}, the bound program holds [{B}, <= {T}] '
                                 '(pass max_text_len= to the Trainer for longer captions)')
            if src.shape[1] < T:
                dst.fill_(fill)
            dst[:, :src.shape[1]].copy_(src, non_blocking=True)
        copy(s['experts'], experts, 'experts')
        copy_text(s['input_ids'], input_ids, pad, 'input_ids'); copy_text(s['attention_mask'], attention_mask, 0, 'attention_mask')
        copy_text(s['labels'], labels, -100, 'labels')
        if (weights is None) != (s['weights'] is None):
            raise ValueError('set_batch: per-sample loss weights must be given for every batch or for none (the loss form is part '
                             'of the captured program)')
        if weights is not None:
            copy(s['weights'], weights.to(F32), 'weights')
            torch.div(s['weights'], s['weights'].shape[0], out=s['dloss'])       # (outside the captured step, like the copies above)

    # ---- input pipeline overlap (round 4): the reference's DataLoader(pin_memory=True) + .to(device, non_blocking=True) hides the
    # host-to-device copy of batch i+1 behind step i (train_caption.py:121-125).  The captured step reads FIXED buffers, so the next batch
    # is staged: prefetch_batch() copies it from (pinned) host memory into a second buffer set on a copy stream, commit_prefetched() moves
    # it into the static buffers with device-to-device copies (56 MB: tens of microseconds) on the compute stream.  Loop shape:
    #     tr.set_batch(first); tr.prefetch_batch(second)
    #     for nxt in loader:  tr.commit_prefetched(); loss = tr.step(); tr.prefetch_batch(*nxt)
    def prefetch_batch(self, experts, input_ids, attention_mask, labels, weights=None):
        if self.static is None:
            raise RuntimeError('prefetch_batch: bind the first batch with set_batch (it fixes the shapes of the captured program)')
        if self._staged:
            raise RuntimeError('prefetch_batch: a staged batch is waiting for commit_prefetched() (one staging set: a second prefetch would overwrite it)')
        if self.staging is None:
            def clone(t):
                return {k: clone(v) for k, v in t.items()} if isinstance(t, dict) else (None if t is None else t.clone())
            self.staging = {k: clone(v) for k, v in self.static.items()}          # (clone, not empty: dloss = 1/B carries over when no weights are bound)
            self.copy_stream = torch.cuda.Stream(device=self.device)
            self._staging_ready, self._staging_free = torch.cuda.Event(), torch.cuda.Event()
            self._staging_free.record(torch.cuda.current_stream())
        # The previous commit must have finished reading the staging set.  The HOST waits for that (an event the compute stream recorded right
        # after the commit's copies), not the copy stream: a host-to-device copy that carries a device-side dependency on the compute stream
        # cost the replayed step 0.45 ms wherever in the step it ran, the same copy without the edge nothing (tools/loader_probe.py,
        # profiles/r6_probe_loader.txt).  Call order that keeps the host from ever idling the device: commit_prefetched() -> step() ->
        # prefetch_batch(next) -- the wait then returns as soon as the device STARTS the step just enqueued.
        self._staging_free.synchronize()

        def on_device(t):
            return any(on_device(v) for v in t.values()) if isinstance(t, dict) else (t is not None and t.is_cuda)
        if any(on_device(t) for t in (experts, input_ids, attention_mask, labels, weights)):
            # device-resident sources were produced on the caller's stream: the copy stream must not read them early (pinned host
            # sources -- the loader contract this path is built for, train_caption.py:121-125 -- need no such edge)
            self.copy_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.copy_stream):
            self._bind(self.staging, experts, input_ids, attention_mask, labels, weights)
            self._staging_ready.record(self.copy_stream)
        self._staged = True

    def commit_prefetched(self):
        """static inputs <- the batch staged by prefetch_batch (enqueued on the current stream, behind the staging copies)"""
        if not self._staged:
            raise RuntimeError('commit_prefetched: nothing is staged (call prefetch_batch first; every staged batch is committed exactly once)')
        cur = torch.cuda.current_stream()
        cur.wait_event(self._staging_ready)

        def move(dst, src):
            if isinstance(dst, dict):
                for k in dst:
                    move(dst[k], src[k])
            elif dst is not None:
                dst.copy_(src, non_blocking=True)
        move(self.static, self.staging)
        self._staging_free.record(cur)
        self._staged = False

    def _snapshot(self):
        """everything a training step mutates: masters (+ bf16 shadows and derived conv shadows follow from them), Adam moments,
        BatchNorm running statistics / counters, the dropout seed, the iteration counter and Python's RNG (instance draws)"""
        return dict(master=[st.master.clone() for st in self.stores], m=[t.clone() for t in self.m], v=[t.clone() for t in self.v],
                    bufs=[b.clone() for mod in (self.enc, self.dec) for b in mod.buffers()], seed=self.seed.clone(), it=self.it,
                    rng=random.getstate(), grads=[st.grad.clone() for st in self.stores], clean=self._grads_clean,
                    pieces={k: (m.clone(), v.clone()) for k, (m, v) in self.piece_state.items()})

    def _restore(self, snap):
        for st, t in zip(self.stores, snap['master']):
            st.master.copy_(t)
            st.refresh()
        for dst, src in zip(self.m + self.v, snap['m'] + snap['v']):
            dst.copy_(src)
        for b, t in zip([b for mod in (self.enc, self.dec) for b in mod.buffers()], snap['bufs']):
            b.copy_(t)
        for st, t in zip(self.stores, snap['grads']):
            st.grad.copy_(t)
        for m, v in self.piece_state.values():                 # rs_ag: moments of pieces first touched during the warm-up start from zero
            m.zero_(); v.zero_()
        for k, (m, v) in snap.get('pieces', {}).items():
            self.piece_state[k][0].copy_(m); self.piece_state[k][1].copy_(v)
        self.seed.copy_(snap['seed'])
        self.it, self._grads_clean = snap['it'], snap['clean']
        random.setstate(snap['rng'])

    def _capture(self):
        """hipGraph capture of the compute segments.  The warm-up passes that precede it (allocator pools, lazily built
        shadows, RCCL channel set-up) run REAL steps, so the training state is snapshotted before and restored after them:
        the first replayed step is then update-for-update the first eager step (one AdamW update per batch like the
        reference loop, train_caption.py:126-135)."""
        snap = self._snapshot()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                           # warm-up outside capture (allocations, lazily built shadows)
            for _ in range(2):
                self._host_prologue()
                with self._wq_scope():                          # (pass 1 counts the writers, pass 2 already overwrites)
                    for seg, coll in self._schedule():
                        seg()
                        if coll is not None:
                            coll()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        if not (self.keep_grads or self.shard or self.rs_ag):
            # every step of this schedule ends in a fused AdamW that leaves the gradient buffers zeroed: the zero_grad fill of the first
            # segment must not be frozen into the graph (it was: two fills, 0.9 GB written = 0.14 ms of every replayed step)
            for st in self.stores:
                st.grad.zero_()
            self._grads_clean = True
            self._graph_no_fill = True                          # the replayed step relies on its predecessor's AdamW for clean buffers
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        # thread_local: RCCL's watchdog thread polls events while we capture (world > 1); only this thread's calls are policed
        mode = dict(capture_error_mode='thread_local')
        graphs = []
        # no cyclic garbage collection while a capture is open: a collection that happens to run inside it finalises whatever cyclic
        # garbage exists at that moment (an earlier Trainer with its CUDAGraphs, events, pinned buffers) with HIP calls that are not
        # legal during capture -- the process aborts ("Fatal Python error: Aborted ... Garbage-collecting", seen once in the GPU suite)
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            with self._wq_scope():
                for seg, coll in self._schedule():
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool, **mode):
                        seg()
                    graphs.append((g, coll))
        finally:
            if gc_was:
                gc.enable()
        self.graphs = graphs

    def step(self):
        """one optimisation step on the bound batch; returns the (device) scalar loss tensor without synchronising."""
        assert self.static is not None, 'call set_batch() first'
        if self.use_graph and self.graphs is None:
            try:
                self._capture()
            except Exception as e:
                # no silent fall-back: a benchmark line must not claim graphs that did not run.  Trainer(allow_eager_fallback=True)
                # continues with eager launches of the same kernels and flips use_graph (callers report tr.use_graph).
                if not self.allow_eager_fallback:
                    raise RuntimeError(f'hipGraph capture of the training step failed ({type(e).__name__}: {e}); pass use_graph=False '
                                       'or allow_eager_fallback=True') from e
                import sys
                print(f'[prismer_amd] hipGraph capture failed ({type(e).__name__}: {e}); continuing with eager launches', file=sys.stderr)
                torch.cuda.synchronize()
                self.graphs, self.use_graph = None, False
                ops.join_side()
        self._host_prologue()
        if self.broadcast_buffers:
            self.sync_buffers()
        if self.exchange is not None:
            self.exchange.begin_step()
        self.trace = []                                         # host enqueue order of compute segments and bucket hand-offs
        if self.use_graph:
            if self._step_open and self._graph_no_fill:
                # the previous replayed step was interrupted between two segments (an exception in a host collective, Ctrl-C): its
                # partly accumulated gradients never reached the AdamW that re-zeroes them, and the graphs carry no fill of their own
                for st in self.stores:
                    st.grad.zero_()
            self._step_open = True
            timed = self.exchange is not None and self.exchange.timing
            def run(coll, ev=None):
                self._issue_event = ev
                if timed:                                       # host time inside the exchange calls: a transport that BLOCKS the host thread (gloo; a
                    t0, w0 = time.perf_counter(), self._host_wait_s                # mis-configured RCCL) stalls the launches behind it -- invisible to device events
                    coll()
                    self._host_comm_s += (time.perf_counter() - t0) - (self._host_wait_s - w0)
                else:
                    coll()
            host_issue = self.exchange_issue == 'host' and self.exchange is not None
            pending = None
            for i, (g, coll) in enumerate(self.graphs):
                self.trace.append(('seg', i))
                g.replay()
                if not host_issue:
                    if coll is not None:
                        run(coll)
                    continue
                # host-driven issue: the event of segment i is recorded now, its host action runs after segment i + 1 has been enqueued (the
                # device never idles while the host wakes up); the LAST host action of the backward (exchange of the front stage + join +
                # sharded tails) has nothing left to hide behind and runs at once
                ev = None
                if coll is not None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())    # (directly behind this segment's replay)
                if pending is not None:
                    run(*pending)
                    pending = None
                if coll is not None:
                    if i + 1 < len(self.graphs) and self.graphs[i + 1][1] is not None:
                        pending = (coll, ev)
                    else:
                        run(coll, ev)
            if timed:
                self._host_comm_steps += 1
            self._step_open = False
        else:
            with self._wq_scope():
                for i, (seg, coll) in enumerate(self._schedule()):
                    self.trace.append(('seg', i))
                    seg()
                    if coll is not None:
                        coll()
        return self.loss_buf

    def state_dict(self):
        """model + optimizer + RNG: what `accelerator.save_state` writes (train_caption.py:174; the reference itself only ever reloads
        the model weights, train_caption.py:103-108).  With shard_optimizer the Adam moments are this rank's slice."""
        return dict(it=self.it, m=[t.clone() for t in self.m], v=[t.clone() for t in self.v],
                    model={k: t.detach().clone() for k, t in self.model.state_dict().items()},      # (state_dict() aliases the live buffers)
                    seed=self.seed.clone(), rng=random.getstate(), world=self.world,
                    shard_optimizer='rs_ag' if self.rs_ag else self.shard,
                    pieces={k: (m.clone(), v.clone()) for k, (m, v) in self.piece_state.items()})       # rs_ag: moments of the owned pieces

    def load_state_dict(self, sd, strict=True):
        """resume: masters (the module parameters are views of the flat buffers), bf16 shadows and derived conv shadows re-derived from
        them, Adam moments, iteration counter (LR schedule position), dropout seed and Python RNG (instance-embedding draws).  Captured
        graphs stay valid: they reference the buffers, not their contents."""
        mine = 'rs_ag' if self.rs_ag else self.shard
        if (sd.get("shard_optimizer", False) or False) != mine or ((self.shard or self.rs_ag) and sd.get("world", 1) != self.world):
            raise ValueError('Trainer.load_state_dict: optimizer sharding of the checkpoint '
                             f"(shard_optimizer={sd.get('shard_optimizer')}, world={sd.get('world')}) differs from this Trainer's")
        if len(sd['m']) != len(self.m) or any(a.shape != b.shape for a, b in zip(sd['m'] + sd['v'], self.m + self.v)):
            raise ValueError('Trainer.load_state_dict: Adam moment buffers do not match this model / freeze configuration')
        self.model.load_state_dict(sd['model'], strict=strict)
        for st in self.stores:
            st.refresh()
        for dst, src in zip(self.m + self.v, list(sd['m']) + list(sd['v'])):
            dst.copy_(src)
        self.piece_state = {k: (m.to(self.device).clone(), v.to(self.device).clone()) for k, (m, v) in sd.get('pieces', {}).items()}
        self.it = int(sd['it'])
        if 'seed' in sd:
            self.seed.copy_(sd['seed'])
        if 'rng' in sd:
            random.setstate(sd['rng'])
        torch.cuda.synchronize(self.device)
