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
Data parallel semantics = DDP's: every rank holds all parameters, gradients are summed over ranks and divided by
world size (folded into the AdamW kernel as grad_scale), BatchNorm uses per-rank batch statistics (no SyncBN in the
reference).  DDP's per-step buffer broadcast from rank 0 is replaced by rank-local BN running statistics
(documented deviation).
"""
import contextlib
import math
import os
import random

import torch

from . import ops
from .dist import bucketed_all_reduce

F32, BF16 = torch.float32, torch.bfloat16


def cosine_lr(it, total, init_lr, min_lr):
    """utils.py:13-17 of the reference."""
    return (init_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * it / total)) + min_lr


class Trainer:
    def __init__(self, model, lr=5e-5, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, min_lr=0.0, total_steps=1000,
                 task='caption', use_graph=True, process_group=None, bucket_mb=64, side_stream=True, micro_batches=1, keep_grads=False):
        self.model = model
        self.enc, self.dec = model.expert_encoder, model.text_decoder
        self.init_lr, self.min_lr, self.total_steps = lr, min_lr, total_steps
        self.wd, self.betas, self.eps = weight_decay, betas, eps
        self.task = task
        self.use_graph = use_graph
        self.pg = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        self.bucket_elems = bucket_mb * 1024 * 1024 // 4
        self.it = 0
        self.graphs = None
        self.static = None
        self.enc.train(); self.dec.train()
        self.enc_prog, self.dec_prog = self.enc._program(), self.dec._program()
        self.stores = [self.enc._store, self.dec._store]
        for st in self.stores:
            st.native_grads = True
            st.managed = True
            st._grad_cur = st.grad
        dev = self.stores[0].master.device
        self.device = dev
        self.m = [torch.zeros(st.n_train, dtype=F32, device=dev) for st in self.stores]
        self.v = [torch.zeros(st.n_train, dtype=F32, device=dev) for st in self.stores]
        self.hyper = torch.zeros(3, dtype=F32, device=dev)
        self.hyper_host = torch.zeros(3, dtype=F32).pin_memory()
        self.table_host = torch.zeros(256, dtype=torch.int32).pin_memory()
        self.table = torch.zeros(256, dtype=torch.int32, device=dev)
        self.seed = self.dec.dropout_seed()
        self.comm_stream = torch.cuda.Stream(device=dev) if self.world > 1 else None
        # Micro-batches (optional, default off): the stems see the whole batch (train-mode BatchNorm statistics), everything
        # after them is sample-independent, so the batch can be cut into `micro_batches` contiguous slices that run as
        # parallel branches (one stream each).  Motivation: at bs32 the phase times shrink only ~0.7x when the batch is
        # halved, i.e. most kernels under-fill the 256 CUs.  Measured on MI355X / ROCm 7.2 (same box, bs32): hipGraph replay
        # 33.2 ms with 1 slice vs 44.7 ms with 2 (this runtime's graph executor handles wide parallel branches poorly;
        # eager launches with 2 slices: 38.5 ms, host-bound) -- so the default stays 1.  Weight-gradient accumulation is
        # race-free either way because every read-modify-write of the gradient buffer lives on the single side stream, and
        # forked streams never re-join work they forked themselves (capture_end crashes on such diamonds here).
        self.micro = micro_batches if (side_stream and micro_batches > 1) else 1
        if side_stream:
            ops.SIDE = ops.SideStream(dev)
            ops.POOL = ops.BranchPool(dev, 3)
            ops.MICRO = ops.BranchPool(dev, self.micro) if self.micro > 1 else ops._NoPool()
        self.loss = None
        self._grads_clean = False
        self.keep_grads = keep_grads           # True: gradients stay readable after step() (tests); False: AdamW zeroes them
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

    def _allreduce_async(self, flat, n, start=0):
        """bucketed SUM all-reduce of flat[start:n] on the communication stream, after everything queued so far on the
        compute stream (those gradients are complete by then)."""
        if self.world == 1 or n <= start:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            bucketed_all_reduce(flat[start:n], n - start, self.bucket_elems, self.pg)

    def _wait_comm(self):
        if self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    # ------------------------------------------------------------------------------------------ step pieces
    def _slices(self, B):
        n = self.micro if (self.micro > 1 and B >= 2 * self.micro) else 1
        cuts = [B * i // n for i in range(n + 1)]
        return [(cuts[i], cuts[i + 1]) for i in range(n)]

    def _seg_forward_dec_backward(self, s):
        # optimizer.zero_grad(): the fused AdamW of the previous step already left the gradient buffers zeroed (they are
        # allocated zeroed); only a step that did not end in _seg_optimizer (first step after an exception) needs the fill
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
        self.sv_t, self.denc, losses = [None] * len(parts), [None] * len(parts), [None] * len(parts)
        dp.kv_prefetch = len(parts) == 1
        for mi, (b0, b1) in enumerate(parts):
            with (ops.MICRO.branch(mi) if len(parts) > 1 else contextlib.nullcontext()):
                Bh = b1 - b0
                enc_out, self.sv_t[mi] = ep.forward_trunk(h[b0 * S:b1 * S], None if xf is None else xf[b0 * Mx:b1 * Mx], Bh, True)
                dp.site_base = mi << 20
                _, loss, sv_d = dp.forward(s['input_ids'][b0:b1], s['attention_mask'][b0:b1], enc_out, s['labels'][b0:b1], self.seed, True)
                if s.get('weights') is not None:               # VQA: (weights * loss).mean()  (prismer_vqa.py:40-41)
                    w = s['weights'][b0:b1].to(F32)
                    dloss, losses[mi] = w / B, (loss * w).sum()
                else:                                          # caption: loss.mean()          (prismer_caption.py:33)
                    dloss, losses[mi] = torch.full((Bh,), 1.0 / B, dtype=F32, device=loss.device), loss.sum()
                self.denc[mi] = dp.backward(sv_d, dloss)       # (valid after the join_side below)
                del sv_d
        dp.site_base = 0
        if len(parts) > 1:
            ops.MICRO.join()
        ops.join_side()
        self.loss_buf = torch.stack(losses).sum() / B

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

    def _seg_enc_backward(self):
        self._seg_enc_trunk_backward()
        self._seg_enc_front_backward()

    def _seg_enc_trunk_backward_joined(self):
        self._seg_enc_trunk_backward()
        ops.join_side()                                        # trunk gradients complete: their all-reduce may start

    def _schedule(self):
        """[(compute segment, collective issued right after it)].  One rank: forward + decoder backward | encoder backward |
        AdamW.  Data parallel: the encoder backward is cut after the trunk, so that the all-reduce of the trunk's gradients
        (adaptors, resampler: the tail of the encoder's flat buffer) overlaps the stems' backward and only the stems' own
        gradients (the head of the buffer) are exchanged on the critical path."""
        enc, dec = self.stores
        if self.world == 1:
            return [(lambda: self._seg_forward_dec_backward(self.static), None), (self._seg_enc_backward, None),
                    (self._seg_optimizer, None)]
        cut = self._trunk_grad_start()
        return [(lambda: self._seg_forward_dec_backward(self.static), lambda: self._allreduce_async(dec.grad, dec.n_train)),
                (self._seg_enc_trunk_backward_joined, lambda: self._allreduce_async(enc.grad, enc.n_train, cut)),
                (self._seg_enc_front_backward, lambda: (self._allreduce_async(enc.grad, cut), self._wait_comm())),
                (self._seg_optimizer, None)]

    def _trunk_grad_start(self):
        """offset in the encoder's flat gradient buffer where the parameters whose gradients are complete after the trunk
        backward begin (transformer.* adaptors, resampler.*, ln_pre / ln_post); everything before it (positional / instance
        embeddings, conv1.* stems) is written by the front backward."""
        st = self.stores[0]
        front = ('positional_embedding', 'instance_embedding', 'conv1.')
        cut, seen_trunk = st.n_train, False
        for n in st.names:
            if not st.is_trainable(n):
                continue
            is_front = n.startswith(front)
            if not is_front and not seen_trunk:
                cut, seen_trunk = st.offset[n], True
            elif is_front and seen_trunk:
                return 0                                       # interleaved layout: no overlap, exchange everything at the end
        return cut

    def _seg_optimizer(self):
        for st, m, v in zip(self.stores, self.m, self.v):
            ops.adamw(st.master, st.grad, m, v, st.shadow, st.n_train, self.hyper, self.betas[0], self.betas[1], self.eps, self.wd,
                      1.0 / self.world, zero_grad=not self.keep_grads)
            st.refresh_derived()
        ops.advance_seed(self.seed)
        self._grads_clean = not self.keep_grads

    def _host_prologue(self):
        """per-step host work: LR schedule (cosine per ITERATION, train_caption.py:127), Adam bias corrections and the
        instance-embedding draw table (vit.py:145-147: Python `random`), shipped with two async pinned copies."""
        self.it += 1
        lr = cosine_lr(self.it - 1, self.total_steps, self.init_lr, self.min_lr)
        self.hyper_host[0] = lr
        self.hyper_host[1] = 1.0 - self.betas[0] ** self.it
        self.hyper_host[2] = 1.0 - self.betas[1] ** self.it
        self.hyper.copy_(self.hyper_host, non_blocking=True)
        if 'obj_detection' in self.enc.experts:
            for i in range(256):
                self.table_host[i] = random.randint(0, 127)
            self.table.copy_(self.table_host, non_blocking=True)

    # ------------------------------------------------------------------------------------------ public API
    def set_batch(self, experts, input_ids, attention_mask, labels, weights=None):
        """(re)binds the static input buffers. First call allocates them; later calls copy into them."""
        for k, v in experts.items():                 # expert-map resolution fixes the program's token geometry
            if k != 'rgb':
                er = (v['label'] if isinstance(v, dict) else v).shape[-1]
                if er != self.enc.expert_resolution:
                    assert self.graphs is None, 'expert resolution changed after graph capture'
                    self.enc.expert_resolution, self.enc._prog = er, None
                    self.enc_prog = self.enc._program()
                break
        if self.static is None:
            def clone(t):
                return {k: clone(v) for k, v in t.items()} if isinstance(t, dict) else t.to(self.device).contiguous().clone()
            self.static = dict(experts=clone(experts), input_ids=input_ids.to(self.device).contiguous().clone(),
                               attention_mask=attention_mask.to(self.device).contiguous().clone(),
                               labels=labels.to(self.device).contiguous().clone(),
                               weights=None if weights is None else weights.to(self.device).contiguous().clone())
            return

        def copy(dst, src):
            if isinstance(dst, dict):
                for k in dst:
                    copy(dst[k], src[k])
            else:
                dst.copy_(src, non_blocking=True)
        s = self.static
        copy(s['experts'], experts); copy(s['input_ids'], input_ids); copy(s['attention_mask'], attention_mask); copy(s['labels'], labels)
        if weights is not None:
            copy(s['weights'], weights)

    def _capture(self):
        s = self.static
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                           # warm-up outside capture (allocations, lazily built shadows)
            for _ in range(2):
                self._host_prologue()
                for seg, coll in self._schedule():
                    seg()
                    if coll is not None:
                        coll()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        # thread_local: RCCL's watchdog thread polls events while we capture (world > 1); only this thread's calls are policed
        mode = dict(capture_error_mode='thread_local')
        graphs = []
        for seg, coll in self._schedule():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, **mode):
                seg()
            graphs.append((g, coll))
        self.graphs = graphs

    def step(self):
        """one optimisation step on the bound batch; returns the (device) scalar loss tensor without synchronising."""
        assert self.static is not None, 'call set_batch() first'
        if self.use_graph and self.graphs is None:
            try:
                self._capture()
            except Exception as e:                      # capture is an optimisation: fall back to eager launches of the same kernels
                import sys
                print(f'[prismer_amd] hipGraph capture failed ({type(e).__name__}: {e}); continuing with eager launches', file=sys.stderr)
                torch.cuda.synchronize()
                self.graphs, self.use_graph = None, False
                ops.join_side()
        self._host_prologue()
        if self.use_graph:
            for g, coll in self.graphs:
                g.replay()
                if coll is not None:
                    coll()
        else:
            for seg, coll in self._schedule():
                seg()
                if coll is not None:
                    coll()
        return self.loss_buf

    def state_dict(self):
        return dict(it=self.it, m=[t.clone() for t in self.m], v=[t.clone() for t in self.v], model=self.model.state_dict())
