#!/usr/bin/env python
"""Headline benchmark: images/sec of Prismer-BASE caption fine-tuning (224^2, 6 experts, batch 32 / GPU, freeze_vision,
T = 30, bf16 storage / fp32 accumulate) on N MI355X -- BASELINE.json `metric`, config[2] (N=1) / config[3] (N=8).

  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = the full training iteration of the reference loop (train_caption.py:126-135): LR schedule, zero_grad, forward
(6 expert stems with train-mode BatchNorm, Experts Resampler, ViT + adaptors, decoder with dropout 0.1, LM head, shifted
label-smoothed CE), backward, gradient all-reduce over RCCL (N > 1) and fused AdamW over the 242.4 M trainable parameters.
Synthetic inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1 only):
  roofline      the bf16 MFMA GEMM kernel family: algorithmic FLOPs (2*M*N*K per launch) / summed launch durations, measured
                with HIP events on the launch stream in an instrumented (eager, non-graph) pass of the same step
  cpu_baseline  the CPU oracle (oracle/prismer_oracle.py, fp32) on a bounded sample of the same workload: batch 8, every logical CPU of
                the host and 32 threads, the better of the two reported with its thread count
  secondary     BASELINE configs 2 and 5, the drop-in nn.Module loop (model -> loss.backward -> torch.optim.AdamW) and a loader-fed
                Trainer (a new pinned-host batch through set_batch every step), all under the same clock as the headline
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GF_PER_IMG = 263.10      # Prismer-BASE, T=30, freeze_vision (3*fwd minus frozen wgrads)
PEAK_TFLOPS = 2500.0           # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
FAMILIES = ['gemm', 'layernorm', 'attention_fwd', 'attention_bwd', 'frontend', 'embed_ce', 'optimizer', 'misc']


def make_inputs(dims, batch, T, seed, device, compact_labels=False):
    """SURVEY 8d synthetic inputs, generated on the device (label experts = uint8 rectangle maps gathered through a
    [256,64] table of std 0.75; dense experts U(-1,1); rgb N(0,1)); text: <s> prompt(3) body </s>, labels mask the prompt."""
    g = torch.Generator(device=device).manual_seed(seed)
    R, E = dims.image_resolution, dims.expert_resolution
    x = {'rgb': torch.randn(batch, 3, R, R, generator=g, device=device)}
    names = [n for n in ['depth', 'normal', 'seg_coco', 'edge', 'obj_detection', 'ocr_detection'] if n in dims.experts or
             (n == 'seg_coco' and 'seg' in dims.experts)]

    def label_map():
        lab = torch.full((batch, E, E), 255, dtype=torch.int64, device=device)
        rect = torch.randint(0, 1 << 20, (batch, 8, 5), generator=g, device=device).cpu()
        for b in range(batch):
            for r in range(8):
                y0, x0 = int(rect[b, r, 0]) % E, int(rect[b, r, 1]) % E
                h, w = 1 + int(rect[b, r, 2]) % (E // 2), 1 + int(rect[b, r, 3]) % (E // 2)
                lab[b, y0:y0 + h, x0:x0 + w] = int(rect[b, r, 4]) % 200
        table = torch.randn(256, 64, generator=g, device=device) * 0.75
        if compact_labels:       # SURVEY 8f #2 input form: uint8 label map + CLIP-feature table, in-painted on the device by the stem
            return lab, {'label_map': lab.to(torch.uint8), 'table': table}
        return lab, table[lab].permute(0, 3, 1, 2).contiguous()
    for n in names:
        if n in ('depth', 'edge'):
            x[n] = torch.rand(batch, 1, E, E, generator=g, device=device) * 2 - 1
        elif n == 'normal':
            x[n] = torch.rand(batch, 3, E, E, generator=g, device=device) * 2 - 1
        else:
            lab, m = label_map()
            if compact_labels:
                x[n] = m
            else:
                x[n] = {'label': m, 'instance': lab.unsqueeze(1)} if n == 'obj_detection' else m
    ids = torch.randint(3, dims.vocab_size, (batch, T), generator=g, device=device)
    ids[:, 0] = 0
    ids[:, 1:4] = torch.tensor([83, 2170, 9], device=device)
    ids[:, T - 1] = 2
    mask = torch.ones(batch, T, dtype=torch.int64, device=device)
    labels = ids.clone()
    labels[:, :4] = -100
    return x, ids, mask, labels


def build_trainer(batch, use_graph, rank, T=30, workload='base_caption', freeze='freeze_vision', compact_labels=False, grad_payload='fp32', issue='device',
                  shard=False):
    from prismer_amd import config as pcfg
    from prismer_amd.model.prismer_caption import PrismerCaption
    from prismer_amd.model.prismer_vqa import PrismerVQA
    from prismer_amd.trainer import Trainer
    torch.manual_seed(0)                                   # identical random-init weights on every rank
    if workload == 'large_vqa':                            # BASELINE config 5: Prismer-LARGE VQA, 480^2, T = 35 + 5
        dims = pcfg.prismer_large()
        cfg = {'experts': pcfg.CAPTION_EXPERTS, 'image_resolution': 480, 'prismer_model': 'prismer_large', 'freeze': 'freeze_vision'}
        model = PrismerVQA(cfg).cuda()
        T = 40
    elif workload == 'base_caption_480':                   # the SHIPPED caption fine-tune resolution (configs/caption.yaml:6,10: 480^2, batch 4 per GPU)
        dims = pcfg.prismer_base(image_resolution=480)
        cfg = {'experts': pcfg.CAPTION_EXPERTS, 'image_resolution': 480, 'prismer_model': 'prismer_base', 'freeze': freeze}
        model = PrismerCaption(cfg).cuda()
    elif workload == 'z_base_caption':                     # BASELINE config 2: PrismerZ-BASE (rgb only, no resampler)
        dims = pcfg.prismerz_base()
        cfg = {'experts': 'none', 'image_resolution': 224, 'prismer_model': 'prismer_base', 'freeze': freeze}
        model = PrismerCaption(cfg).cuda()
    else:
        dims = pcfg.prismer_base()
        cfg = {'experts': pcfg.CAPTION_EXPERTS, 'image_resolution': 224, 'prismer_model': 'prismer_base', 'freeze': freeze}
        model = PrismerCaption(cfg).cuda()
    tr = Trainer(model, lr=5e-5, weight_decay=0.05, total_steps=10000, task='caption', use_graph=use_graph, grad_payload=grad_payload,
                 shard_optimizer=shard, exchange_issue=issue)
    x, ids, mask, labels = make_inputs(dims, batch, T, 1234 + rank, torch.device('cuda'), compact_labels)
    weights = None
    if workload == 'large_vqa':
        labels[:, :35] = -100                              # only the answer span is scored (prismer_vqa.py:32-33)
        weights = torch.rand(batch, device='cuda') * 0.8 + 0.2
    tr.set_batch(x, ids, mask, labels, weights)
    n_train = sum(st.n_train for st in tr.stores)
    return tr, dims, n_train


def kernel_family_pass(tr, steps):
    """instrumented eager pass: per-family HIP-event timing through the library's measurement hooks."""
    from prismer_amd._lib import lib
    from prismer_amd import ops
    tr.use_graph = False
    ops.join_side(); ops.SIDE = None           # single stream: per-launch durations without cross-stream contention
    tr.step(); torch.cuda.synchronize()
    # decoder share of the step (round-4 review): HIP events around the decoder program's forward / backward entry points
    dec_ev = []
    dp = tr.dec_prog
    saved = {n: getattr(dp, n) for n in ('forward', 'backward_start', 'backward_layers', 'backward_finish')}

    def timed(fn):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            dec_ev.append((e0, e1))
            return r
        return wrapper
    for n, fn in saved.items():
        setattr(dp, n, timed(fn))
    lib.ph_prof_enable(1)
    try:
        for _ in range(steps):
            tr.step()
    except BaseException:
        lib.ph_prof_enable(0)                               # (a failed instrumented step must not leave the profiler on for the later legs)
        raise
    finally:
        for n in saved:
            delattr(dp, n)                                  # (instance attributes shadowing the methods)
    import tempfile
    dump = os.environ.get('PH_PROF_DUMP') or os.path.join(tempfile.gettempdir(), f'ph_prof_{os.getpid()}.csv')
    lib.ph_prof_dump(dump.encode())
    out = (ctypes.c_double * (len(FAMILIES) * 4))()
    lib.ph_prof_collect(out)
    lib.ph_prof_enable(0)
    fam = {}
    for i, name in enumerate(FAMILIES):
        ms, fl, by, n = out[4 * i:4 * i + 4]
        fam[name] = dict(ms_per_step=ms / steps, tflop_per_step=fl / steps / 1e12, launches_per_step=n / steps, gbytes_per_step=by / steps / 1e9)
    # GEMM launches by KERNEL class (the library tags every profiled GEMM call with the class its dispatch chose)
    names = ['gemm_kernel<128,*> (128x128 / 128x64 register-staged)', 'gemm_kernel<64,64> (64x64 register-staged)',
             'gemm_ks2_kernel (64x64, k loop split inside the block: the decoder\'s M = 960 launches)',
             'big::gemm_big_kernel<28,*,*> (256x128 LDS-DMA ping-pong, single launch)',
             'big::gemm_big_grouped_kernel / gemm_big_conv_kernel (256x128 LDS-DMA, grouped: weight gradients, stem convolutions)',
             'gemm_grouped_kernel (register-staged, grouped)', 'splitk_reduce']
    cls = {}
    try:
        for line in open(dump):
            f = line.rstrip('\n').split(',')
            if f[0] == '0' and len(f) >= 5 and f[-1].lstrip('-').isdigit() and int(f[-1]) >= 0:
                c = cls.setdefault(int(f[-1]), [0.0, 0.0, 0])
                c[0] += float(f[1]); c[1] += float(f[2]); c[2] += 1
    except OSError:
        pass
    if not os.environ.get('PH_PROF_DUMP') and os.path.isfile(dump):
        os.remove(dump)
    fam['_gemm_classes'] = {names[k] if k < len(names) else str(k): dict(ms_per_step=v[0] / steps, tflop_per_step=v[1] / steps / 1e12, launches_per_step=v[2] / steps)
                            for k, v in cls.items()}
    fam['_decoder_ms_per_step'] = sum(a.elapsed_time(b) for a, b in dec_ev) / steps
    return fam


def pmc_traffic(live_launches=None):
    """HBM bytes per GEMM launch from the committed PMC passes (profiles/rN_pmc_gemm.json, newest round first: rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate runs of this bench, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md; tools/profile_round4.sh is the
    recipe); counters cannot be read from inside the process, so this is (None, None) when the file is absent.  Returns the
    per-launch bytes and the launch count the passes saw, so that a stale file shows next to the live launch count."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_gemm.json')), key=lambda f: int(re.search(r'r(\d+)', os.path.basename(f)).group(1)), reverse=True)
    for p in files[:1]:                                    # the NEWEST committed pass only: no silent fall-back through older rounds (round-5 review)
        d = json.load(open(p))
        if live_launches is not None and abs(d['launches_per_step'] - live_launches) > 0.5:
            return None, dict(file='profiles/' + os.path.basename(p), launches_per_step=d['launches_per_step'], refused=f'stale: the live step has {live_launches:g} GEMM launches')
        return round(d['hbm_bytes_per_launch']), dict(file='profiles/' + os.path.basename(p), launches_per_step=d['launches_per_step'],
                                                      whole_step_hbm_gb=round(d['whole_step_hbm_gb'], 1))
    return None, None


def cpu_baseline(seconds_budget=30.0, only_cores=None):
    """CPU oracle (a port: plain-PyTorch fp32 restatement of the reference modules, pinned to reference outputs by
    tests/golden) timed on this host: Prismer-BASE caption train step (fwd+bwd+AdamW), batch 8, T=30, freeze_vision -- the batch
    BASELINE.md section 3 times the reference classes at.  Two thread counts inside the budget: every logical CPU of the host (what
    "the same host" means) and 32 (torch's CPU kernels often lose to their own synchronisation beyond that on these op sizes);
    `value` is the better one and `cores` says which, both are in `sample`."""
    from oracle import prismer_oracle as O
    from prismer_amd import config as pcfg, synth
    d = pcfg.prismer_base()
    B, T = 8, 30
    esd, dsd = synth.synth_encoder_state(d, 0), synth.synth_decoder_state(d, 0)
    names = ['expert_encoder.' + k for k in esd] + ['text_decoder.' + k for k in dsd]
    fm = O.freeze_mask(names, 'freeze_vision')
    leaves, seen = [], set()
    for pre, sd in (('expert_encoder.', esd), ('text_decoder.', dsd)):
        for k, v in sd.items():
            if v.is_floating_point() and 'running' not in k and id(v) not in seen and fm[pre + k]:
                seen.add(id(v)); v.requires_grad_(True); leaves.append(v)
    opt = torch.optim.AdamW(leaves, lr=5e-5, weight_decay=0.05)
    x = synth.synth_experts(d, B, seed=1)
    ids, mask, labels = synth.synth_text(d, B, T, seed=1)
    tab = list(range(128)) * 2

    def step():
        opt.zero_grad()
        loss, _, _ = O.caption_loss(esd, dsd, x, ids, mask, labels, d, train_bn=True, instance_table=tab, bn_updates={})
        loss.backward()
        opt.step()
    ncpu = os.cpu_count() or 1

    def timed(cores, max_steps, budget):
        torch.set_num_threads(cores)
        times, t_start = [], time.time()
        while len(times) < max_steps and (not times or time.time() - t_start + min(times) < budget):
            t0 = time.time(); step(); times.append(time.time() - t0)
        return min(times), len(times)                      # the first step carries one-off allocation cost: the best step counts
    if only_cores is not None:                             # child process of the all-cores attempt: one bounded measurement, one JSON line
        t, n = timed(only_cores, 2, seconds_budget)
        print(json.dumps({'t': t, 'n': n}), flush=True)
        return None
    results, note = {}, ''
    base = min(32, ncpu)
    results[base] = timed(base, 3, seconds_budget * 0.6)
    if ncpu > base:
        # every logical CPU: in a child process with a hard time limit -- torch's CPU kernels can collapse with hundreds of threads on these
        # op sizes (one step then takes minutes), and a running step cannot be interrupted from inside the process
        import subprocess
        try:
            r = subprocess.run([sys.executable, '-c', f'import bench; bench.cpu_baseline({seconds_budget * 0.4}, only_cores={ncpu})'], cwd=ROOT,
                               capture_output=True, text=True, timeout=seconds_budget * 0.4 + 15)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            results[ncpu] = (d['t'], d['n'])
        except Exception as e:
            note = f'; {ncpu} threads: no step finished inside {seconds_budget * 0.4 + 15:.0f} s ({type(e).__name__}): slower than {base} threads'
    best = min(results, key=lambda c: results[c][0])
    dt = results[best][0]
    return dict(value=round(B / dt, 3), unit='images/sec', cores=best, kind='port',
                sample=f'Prismer-BASE caption train step (fwd+bwd+AdamW, freeze_vision, fp32 oracle), batch {B}, T={T}; ' +
                       '; '.join(f'{c} threads: best of {n} step(s) = {t:.2f} s = {B / t:.2f} images/s' for c, (t, n) in sorted(results.items())) +
                       note + f' (host has {ncpu} logical cpus)')


def dropin_leg(steps=10, warmup=3, batch=32, fused=False):
    """What the reference's own loop gets from the drop-in modules (train_caption.py:121-136): `loss = model(experts, caption, prefix=...)`,
    `loss.backward()`, `torch.optim.AdamW.step()` -- each top module one torch.autograd.Function over the HIP layer programs, parameter
    gradients handed to autograd, PyTorch's optimizer on the fp32 masters (bf16 shadows refreshed on the next forward).  No hipGraph."""
    from prismer_amd import config as pcfg
    from prismer_amd.model.prismer_caption import PrismerCaption
    torch.manual_seed(0)
    dims = pcfg.prismer_base()
    model = PrismerCaption({'experts': pcfg.CAPTION_EXPERTS, 'image_resolution': 224, 'prismer_model': 'prismer_base', 'freeze': 'freeze_vision'}).cuda()
    model.train()
    from prismer_amd.optim import AdamW as FusedAdamW            # (same interface; one launch per parameter store instead of torch's multi-tensor passes)
    opt = (FusedAdamW if fused else torch.optim.AdamW)([p for p in model.parameters() if p.requires_grad], lr=5e-5, weight_decay=0.05)
    x, ids, mask, _ = make_inputs(dims, batch, 30, 1234, torch.device('cuda'))
    caption = {'input_ids': ids, 'attention_mask': mask}

    def step():
        loss = model(x, caption, prefix=4)                 # (token ids instead of strings: no RoBERTa vocabulary on an offline box)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    first = step().item()
    for _ in range(warmup - 1):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    check_loss('dropin', first, loss.item())
    if fused:
        assert opt.fused_launches >= 1 and opt.plain_updates == 0, 'the fused optimizer fell back to per-tensor updates'
    return dict(value=round(batch * steps / dt, 2), unit='images/sec', batch=batch, steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 3),
                hip_graph=False, first_loss=round(float(first), 4), final_loss=round(float(loss.item()), 4),
                config='Prismer-BASE caption fine-tune through the drop-in nn.Modules: model(experts, caption) -> loss.backward() -> ' +
                       ('prismer_amd.optim.AdamW' if fused else 'torch.optim.AdamW') + ' (reference loop train_caption.py:126-135), eager launches')


def loader_leg(steps=10, warmup=3, batch=32):
    """The native Trainer fed like a real loader feeds it: every step a NEW batch arrives in pinned host memory; it is staged on a copy stream
    while the previous step runs (Trainer.prefetch_batch) and moved into the static buffers device-to-device before the replayed step -- compact label experts (uint8 maps + feature
    tables, in-painted on the device: 56 MB per step instead of 1.2 GB of dense 64-channel fp32 maps)."""
    tr, dims, _ = build_trainer(batch, True, 0, compact_labels=True)

    def pin(t):
        return {k: pin(v) for k, v in t.items()} if isinstance(t, dict) else t.cpu().pin_memory()
    batches = []
    for i in range(3):                                      # three distinct host batches, cycled
        x, ids, mask, labels = make_inputs(dims, batch, 30, 4321 + i, torch.device('cuda'), True)
        batches.append((pin(x), ids.cpu().pin_memory(), mask.cpu().pin_memory(), labels.cpu().pin_memory()))
    nbytes = sum(t.numel() * t.element_size() for t in _leaves(batches[0][0])) + sum(t.numel() * t.element_size() for t in batches[0][1:])
    tr.set_batch(*batches[0])
    first = tr.step().item()
    tr.prefetch_batch(*batches[1])
    for i in range(1, warmup):
        tr.commit_prefetched(); tr.step(); tr.prefetch_batch(*batches[(i + 1) % 3])
    torch.cuda.synchronize()
    losses = []
    t0 = time.perf_counter()
    for i in range(steps):                                  # batch i+1 travels over PCIe while step i runs (Trainer.prefetch_batch)
        tr.commit_prefetched()
        loss = tr.step()
        losses.append(loss.clone())                         # (device-side copy: the loss is only read after the timed loop)
        tr.prefetch_batch(*batches[(warmup + i + 1) % 3])   # (waits on the HOST until the device has started the step just enqueued: see Trainer.prefetch_batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for l in losses:
        check_loss('loader', first, l.item())               # EVERY timed step, not only the last one
    return dict(value=round(batch * steps / dt, 2), unit='images/sec', batch=batch, steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 3),
                h2d_mbytes_per_step=round(nbytes / 1e6, 1), hip_graph=bool(tr.graphs is not None), first_loss=round(float(first), 4),
                final_loss=round(float(loss.item()), 4),
                config='Prismer-BASE caption fine-tune, native Trainer, a new batch from pinned host memory every step (prefetch_batch on a copy stream + commit_prefetched + step), '
                       'compact label experts in-painted on the device')


def check_loss(leg, first, last):
    """A leg whose loss is not a sane training loss must not publish a throughput (round 4: the loader leg reported 1e8..1e18 and nobody
    looked).  Sane = finite, positive and below twice the loss of the leg's FIRST step (synthetic batches: the loss falls or stays)."""
    first, last = float(first), float(last)
    if not (math.isfinite(first) and math.isfinite(last)) or last <= 0.0 or last >= 2.0 * max(first, 1e-6):
        raise RuntimeError(f'{leg}: loss check failed (first step {first:.6g}, last step {last:.6g}): the timed steps did not run on valid data')


def _leaves(t):
    if isinstance(t, dict):
        for v in t.values():
            yield from _leaves(v)
    else:
        yield t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--freeze', default='freeze_vision', choices=['freeze_vision', 'none'],
                    help='freeze_vision = the shipped fine-tune setting (headline); none = all parameters trainable (secondary)')
    ap.add_argument('--workload', default='base_caption', choices=['base_caption', 'large_vqa', 'z_base_caption'],
                    help='base_caption = the headline metric (BASELINE config 3/4); large_vqa = config 5 (secondary; use --batch 16)')
    ap.add_argument('--compact-labels', action='store_true',
                    help='label experts as uint8 maps + feature tables, in-painted on the device (SURVEY 8f #2; secondary: the headline '
                         'keeps the reference loader contract of dense 64-channel fp32 maps)')
    ap.add_argument('--grad-payload', default='auto', choices=['auto', 'fp32', 'bf16'],
                    help='N > 1: gradient exchange payload. fp32 = DDP semantics; bf16 = pre-scaled by 1/world, half the bytes; auto (default) = fp32 unless a '
                         'probe all-reduce at start-up measures < 150 GB/s of bus bandwidth on >= 4 ranks (the decision is printed in grad_exchange)')
    ap.add_argument('--issue', default='device', choices=['device', 'host'],
                    help="N > 1: how a finished backward stage reaches the communication stream: 'device' = stream event edge (default); 'host' = the next "
                         "segment is enqueued, then the host waits for the stage and launches the collectives (no device-side edge: see Trainer.exchange_issue)")
    ap.add_argument('--shard', default='none', choices=['none', 'zero1', 'rs_ag'],
                    help='N > 1: none = replicated AdamW behind an all-reduce; zero1 = sharded AdamW + broadcasts; rs_ag = reduce-scatter + '
                         'sharded AdamW + all-gather (FSDP SHARD_GRAD_OP pattern)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary workloads (BASELINE configs 2 and 5) of the N=1 line')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
        # 127.0.0.1) through torch.distributed.run and hand its exit code back; never degrade to a 1-rank run.
        import socket
        import subprocess
        one_dev = os.environ.get('PRISMER_ONE_DEVICE', '0') != '0'
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus and not one_dev:
            raise SystemExit(f'bench.py: --gpus {args.gpus} requested but {n_dev} GPU(s) visible; refusing to run fewer ranks')
        sock = socket.socket(); sock.bind(('127.0.0.1', 0)); port = sock.getsockname()[1]; sock.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback path exists)')
    # PRISMER_DIST_BACKEND=gloo + PRISMER_ONE_DEVICE=1: dry run of the multi-rank path on a single GPU (RCCL refuses two ranks
    # per device); the real launch uses the defaults: one GPU per rank, backend nccl (= RCCL)
    backend = os.environ.get('PRISMER_DIST_BACKEND', 'nccl')
    if os.environ.get('PRISMER_ONE_DEVICE', '0') != '0':
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            torch.distributed.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)')
    ranks_seen = 1
    if world > 1:                                          # proof that the collective transport spans all ranks
        t = torch.ones(1, device='cuda')
        torch.distributed.all_reduce(t)
        ranks_seen = int(t.item())
        assert ranks_seen == world, (ranks_seen, world)

    tr, dims, n_train = build_trainer(args.batch, not args.no_graph, rank, workload=args.workload, freeze=args.freeze,
                                      compact_labels=args.compact_labels, grad_payload=args.grad_payload, issue=args.issue,
                                      shard={'none': False, 'zero1': True, 'rs_ag': 'rs_ag'}[args.shard])
    # algorithmic train GFLOP per image (SURVEY 8d / BASELINE.md section 2): (freeze_vision, none)
    gf_img = {'base_caption': (TRAIN_GF_PER_IMG, 307.26), 'z_base_caption': (134.30, 167.59), 'large_vqa': (2987.8, 3724.7)}[args.workload][
        0 if args.freeze == 'freeze_vision' else 1]
    headline = args.workload == 'base_caption' and args.freeze == 'freeze_vision' and not args.compact_labels

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    first_loss = None
    for i in range(args.warmup):
        l = tr.step()
        if i == 0:
            first_loss = float(l.item())
    barrier()
    if world > 1 and tr.exchange is not None:
        tr.exchange.timing = True                          # events on the communication stream + around the join (a few us per step)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt
    comm = tr.comm_timing() if world > 1 else None
    if comm is not None:                                   # diagnosis of the first multi-GPU run: how much of the exchange is hidden
        comm['rccl_env'] = {k: os.environ[k] for k in ('NCCL_ALGO', 'NCCL_PROTO', 'NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS', 'NCCL_NCHANNELS_PER_PEER',
                                                        'RCCL_MSCCL_ENABLE', 'HSA_ENABLE_IPC_MODE_LEGACY') if k in os.environ}
    final_loss = float(loss.item())
    if first_loss is not None:
        check_loss('headline', first_loss, final_loss)       # raises: no line is printed for a step that did not compute a sane loss
    if not args.no_graph and not (tr.use_graph and tr.graphs is not None):
        raise SystemExit('bench.py: hipGraph replay was requested but the Trainer is running eager launches')

    out = {
        'metric': {'base_caption': 'images/sec Prismer-BASE caption train, 224^2 + 6 experts, bs32/GPU',
                   'z_base_caption': 'images/sec PrismerZ-BASE caption train, 224^2, rgb only, bs32/GPU',
                   'large_vqa': 'images/sec Prismer-LARGE VQAv2 fine-tune, 480^2 + 6 experts, bs16/GPU'}[args.workload] +
                  ('' if args.freeze == 'freeze_vision' else ' (freeze: none)') + (' (compact label inputs)' if args.compact_labels else ''),
        'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
        'data': 'synthetic',
        'config': {'workload': ('Prismer-BASE caption fine-tune step (fwd+bwd+allreduce+AdamW), 224^2, 6 experts + Resampler, T=30, '
                                'freeze_vision, dropout 0.1, train-mode BatchNorm') if headline else
                               f'{args.workload} fine-tune step (fwd+bwd+allreduce+AdamW), freeze={args.freeze}, dropout 0.1' if args.workload != 'large_vqa' else
                               ('Prismer-LARGE VQAv2 fine-tune step (fwd+bwd+allreduce+AdamW), 480^2, 6 experts + Resampler, T=35+5, '
                                'freeze_vision, weighted loss, dropout 0.1, train-mode BatchNorm'),
                   'model': {'base_caption': 'prismer_base', 'z_base_caption': 'prismerz_base', 'large_vqa': 'prismer_large (VQA, 480^2, T=40)'}[args.workload], 'global_batch': world * args.batch, 'seq_len': dims.seq_len, 'text_len': 30 if args.workload == 'base_caption' else 40,
                   'parallelism': f'dp{world}', 'ranks_in_collective': ranks_seen, 'collective_backend': ('rccl' if backend == 'nccl' else backend) if world > 1 else None,
                   'grad_exchange': dict(tr.exchange_desc() or {}, **(comm or {})) if world > 1 else None,
                   'trainable_params': n_train, 'hip_graph': bool(tr.use_graph and tr.graphs is not None),
                   'first_loss': None if first_loss is None else round(first_loss, 4), 'final_loss': round(final_loss, 4)},
        'step_tflops': round(value / world * gf_img / 1e3, 2),
        'step_mfma_frac': round(value / world * gf_img / 1e3 / PEAK_TFLOPS, 4),
    }
    if rank == 0 and world == 1 and not headline:
        args.no_cpu_baseline = True                        # the CPU leg is defined on the headline workload only
    if rank == 0 and world == 1:
        if not args.no_roofline:
            fam = kernel_family_pass(tr, min(args.steps, 5))
            gemm_classes, decoder_ms = fam.pop('_gemm_classes'), fam.pop('_decoder_ms_per_step')
            g = fam['gemm']
            ach = g['tflop_per_step'] / (g['ms_per_step'] / 1e3) if g['ms_per_step'] > 0 else 0.0
            out['roofline'] = {'bound': 'mfma', 'kernel': 'gemm_*kernel<*> (all bf16 MFMA GEMM launches of one step: gemm_kernel, gemm_ks2_kernel, gemm_grouped_kernel, big::gemm_big_kernel, big::gemm_big_grouped_kernel)',
                               'achieved': round(ach, 1), 'peak': PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_TFLOPS, 4),
                               'traffic': pmc_traffic(g['launches_per_step'])[0], 'traffic_source': pmc_traffic(g['launches_per_step'])[1], 'algorithmic_bytes_per_launch': round(g['gbytes_per_step'] * 1e9 / max(g['launches_per_step'], 1)),
                               'avg_launch_us': round(g['ms_per_step'] * 1e3 / max(g['launches_per_step'], 1), 2),
                               'launches_per_step': g['launches_per_step'], 'gemm_tflop_per_step': round(g['tflop_per_step'], 3)}
            if gemm_classes:                                # the single most expensive GEMM kernel of the step, beside the family figure
                dn, dv = max(gemm_classes.items(), key=lambda kv: kv[1]['ms_per_step'])
                dach = dv['tflop_per_step'] / (dv['ms_per_step'] / 1e3) if dv['ms_per_step'] > 0 else 0.0
                out['roofline']['dominant_kernel'] = {'name': dn, 'launches_per_step': round(dv['launches_per_step'], 1), 'ms_per_step': round(dv['ms_per_step'], 3),
                                                      'achieved': round(dach, 1), 'frac': round(dach / PEAK_TFLOPS, 4),
                                                      'note': 'algorithmic FLOPs / HIP-event time of the instrumented eager pass (2-3 us of event overhead per launch included)'}
                out['roofline']['gemm_kernel_classes'] = {k: {'launches_per_step': round(v['launches_per_step'], 1), 'ms_per_step': round(v['ms_per_step'], 3),
                                                              'frac': round((v['tflop_per_step'] / (v['ms_per_step'] / 1e3) if v['ms_per_step'] > 0 else 0.0) / PEAK_TFLOPS, 4)}
                                                          for k, v in sorted(gemm_classes.items(), key=lambda kv: -kv[1]['ms_per_step'])}
            out['kernel_families_ms_per_step'] = {k: round(v['ms_per_step'], 3) for k, v in fam.items()}
            out['decoder_ms_per_step'] = round(decoder_ms, 3)          # decoder forward (+ LM head, CE) + decoder backward, eager instrumented pass
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        if headline and not args.no_secondary and not args.no_graph:
            # BASELINE configs 2 and 5 on one GPU, under the same clock as the headline (round 3): 10 timed steps each after 3 warm-up
            # steps, hipGraph replay, same synthetic-input recipe.  Reported beside the headline so that a dispatch tuned on one
            # workload cannot silently regress another.
            del tr
            import gc
            gc.collect(); torch.cuda.empty_cache()
            sec = {}
            for wl, bs, gf in (('z_base_caption', 32, 134.30), ('large_vqa', 16, 2987.8), ('base_caption_480', 4, None)):
                try:
                    tr2, d2, _ = build_trainer(bs, True, 0, workload=wl)
                    f2 = tr2.step().item()
                    for _ in range(2):
                        tr2.step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(10):
                        l2 = tr2.step()
                    torch.cuda.synchronize()
                    dt2 = time.perf_counter() - t1
                    check_loss(wl, f2, l2.item())
                    v2 = bs * 10 / dt2
                    sec[wl] = dict(value=round(v2, 2), unit='images/sec', batch=bs, steps=10, warmup=3, ms_per_step=round(dt2 / 10 * 1e3, 3),
                                   step_mfma_frac=None if gf is None else round(v2 * gf / 1e3 / PEAK_TFLOPS, 4), hip_graph=bool(tr2.graphs is not None),
                                   first_loss=round(float(f2), 4), final_loss=round(float(l2.item()), 4),
                                   config={'z_base_caption': 'PrismerZ-BASE caption fine-tune, 224^2, rgb only (BASELINE config 2)',
                                           'large_vqa': 'Prismer-LARGE VQAv2 fine-tune, 480^2, 6 experts, T=35+5, weighted loss (BASELINE config 5, one GPU)',
                                           'base_caption_480': 'Prismer-BASE caption fine-tune at the SHIPPED resolution and per-GPU batch (configs/caption.yaml:6,10: '
                                                               '480^2 rgb = 900 + 64 tokens, bicubic position re-grid, batch 4), T=30'}[wl])
                    del tr2
                    gc.collect(); torch.cuda.empty_cache()
                except Exception as e:                     # a secondary leg must never take the headline line down
                    sec[wl] = dict(error=f'{type(e).__name__}: {e}'[:300])
            # round 4: the two boundaries the library advertises besides the native Trainer, under the same clock
            # round 6: 'dropin_fused_adamw' = the same reference loop with prismer_amd.optim.AdamW in place of torch.optim.AdamW
            for name, leg in (('dropin', dropin_leg), ('dropin_fused_adamw', lambda: dropin_leg(fused=True)), ('loader', loader_leg)):
                try:
                    sec[name] = leg()
                except Exception as e:
                    sec[name] = dict(error=f'{type(e).__name__}: {e}'[:300])
                gc.collect(); torch.cuda.empty_cache()
            out['secondary'] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
