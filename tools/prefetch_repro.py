"""Round-5 diagnosis of the loader leg (prefetch_batch / commit_prefetched under hipGraph replay computed garbage after other
Trainers had lived in the process: VERDICT round 4, weak #1).

    python tools/prefetch_repro.py [--prior all|none] [--poison] [--steps N] [--out FILE]

What it does, in ONE process like bench.py:
  1. (--prior all) builds, steps and deletes the Trainers bench.py runs before the loader leg (headline, PrismerZ, LARGE-VQA, drop-in).
  2. builds the loader Trainer and records, DURING CAPTURE, every device pointer handed to the library (Tensor.data_ptr hook); after
     capture each recorded storage is looked up in torch.cuda.memory_snapshot(): a storage whose block is no longer 'active_allocated'
     is a pointer the replayed graph still uses although the caching allocator is free to hand the block to somebody else.
  3. (--poison) soaks up the allocator's free general-pool blocks with 0xA5-filled tensors, replays three steps and reports which of
     them changed = memory the graph writes without owning it.
  4. runs the loader loop exactly like bench.loader_leg (no host synchronisation), logging per step a checksum of every static input leaf
     after commit_prefetched and the loss; compares with the checksums of the host batches and with the set_batch-per-step trajectory
     of an identically initialised Trainer.
"""
import argparse
import gc
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

REC = None
_orig_data_ptr = torch._C.TensorBase.data_ptr


def _data_ptr(self):
    p = _orig_data_ptr(self)
    if REC is not None and self.is_cuda and torch.cuda.is_current_stream_capturing():
        st = self.untyped_storage()
        f = sys._getframe(1)
        REC.append((st.data_ptr(), st.nbytes(), f.f_code.co_name, f.f_back.f_code.co_name if f.f_back else '',
                    f.f_back.f_back.f_code.co_name if f.f_back and f.f_back.f_back else ''))
    return p


def blocks():
    out = []
    for seg in torch.cuda.memory_snapshot():
        a = seg['address']
        for b in seg['blocks']:
            addr = b.get('address', a)
            out.append((addr, b['size'], b['state'], tuple(seg.get('segment_pool_id', (0, 0))), seg['stream']))
            a = addr + b['size']
    out.sort()
    return out


def lookup(bl, p):
    import bisect
    i = bisect.bisect_right(bl, (p, 1 << 62)) - 1
    if i >= 0 and bl[i][0] <= p < bl[i][0] + bl[i][1]:
        return bl[i]
    return None


def leaves(t, prefix=''):
    if isinstance(t, dict):
        for k, v in t.items():
            yield from leaves(v, f'{prefix}{k}.')
    elif t is not None:
        yield prefix[:-1], t


def checksum(t):
    return t.reshape(-1).view(torch.uint8).to(torch.int64).sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--prior', default='all')
    ap.add_argument('--poison', action='store_true')
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--out', default='')
    ap.add_argument('--health', action='store_true', help='synchronise and log state health (loss, hyper, seed, |master|, |m|, |v|) around every step')
    ap.add_argument('--no-prefetch', action='store_true', help='after the poison phase feed the loop with set_batch instead of prefetch/commit')
    ap.add_argument('--poison-byte', type=lambda v: int(v, 0), default=0xA5)
    args = ap.parse_args()
    global REC
    rep = dict(prior=args.prior, poison=args.poison)
    if args.prior == 'all':
        for wl, bs in (('base_caption', 32), ('z_base_caption', 32), ('large_vqa', 16)):
            tr, _, _ = bench.build_trainer(bs, True, 0, workload=wl)
            for _ in range(4):
                loss = tr.step()
            rep[f'prior_{wl}_loss'] = float(loss.item())
            del tr
            gc.collect(); torch.cuda.empty_cache()
        rep['prior_dropin'] = bench.dropin_leg(steps=3, warmup=2)['final_loss']
        gc.collect(); torch.cuda.empty_cache()

    batch = 32
    tr, dims, _ = bench.build_trainer(batch, True, 0, compact_labels=True)
    health_log = []

    def health(tag, loss=None):
        if not args.health:
            return
        torch.cuda.synchronize()
        h = dict(tag=tag, loss=None if loss is None else float(loss.item()), hyper=[float(v) for v in tr.hyper.cpu()], seed=int(tr.seed.cpu()[0]), it=tr.it,
                 master=[float(st.master.abs().max()) for st in tr.stores], m=[float(t.abs().max()) for t in tr.m], v=[float(t.abs().max()) for t in tr.v],
                 grad=[float(st.grad.abs().max()) for st in tr.stores], shadow=[float(st.shadow.float().abs().max()) for st in tr.stores],
                 static={k: float(t.float().abs().max()) for k, t in leaves(tr.static)})
        if tr.staging is not None:
            h['staging'] = {k: float(t.float().abs().max()) for k, t in leaves(tr.staging)}
        health_log.append(h)

    def pin(t):
        return {k: pin(v) for k, v in t.items()} if isinstance(t, dict) else t.cpu().pin_memory()
    batches = []
    for i in range(3):
        x, ids, mask, labels = bench.make_inputs(dims, batch, 30, 4321 + i, torch.device('cuda'), True)
        batches.append((pin(x), ids.cpu().pin_memory(), mask.cpu().pin_memory(), labels.cpu().pin_memory()))
    expect = []
    for b in batches:
        d = dict(experts=b[0], input_ids=b[1], attention_mask=b[2], labels=b[3])
        expect.append({k: int(checksum(v)) for k, v in leaves(d)})
    tr.set_batch(*batches[0])
    torch.Tensor.data_ptr = _data_ptr
    REC = []
    loss0 = tr.step()                                        # warm-up passes + capture + first replay
    rec, REC = REC, None
    torch.Tensor.data_ptr = _orig_data_ptr
    torch.cuda.synchronize()
    rep['loss_step1'] = float(loss0.item())
    gc.collect()
    bl = blocks()
    dangling = {}
    for sp, nb, f1, f2, f3 in rec:
        b = lookup(bl, sp)
        state = 'unmapped' if b is None else b[2]
        if state != 'active_allocated':
            key = (f1, f2, f3, state, None if b is None else b[3])
            e = dangling.setdefault(key, dict(n=0, bytes=0, example=hex(sp)))
            e['n'] += 1; e['bytes'] = max(e['bytes'], nb)
    rep['recorded_pointers'] = len(rec)
    rep['dangling'] = [dict(site='/'.join(k[:3]), state=k[3], pool=str(k[4]), **v) for k, v in sorted(dangling.items(), key=lambda kv: -kv[1]['n'])]
    rep['free_general_blocks_mb'] = round(sum(b[1] for b in bl if b[2] == 'inactive' and b[3] == (0, 0)) / 1e6, 1)
    rep['free_private_blocks_mb'] = round(sum(b[1] for b in bl if b[2] == 'inactive' and b[3] != (0, 0)) / 1e6, 1)

    if args.poison:
        free = sorted([b for b in bl if b[2] == 'inactive' and b[3] == (0, 0) and b[1] >= (1 << 16)], key=lambda b: -b[1])
        poison = []
        for b in free[:400]:
            try:
                t = torch.empty(b[1] - 512 if b[1] > 4096 else b[1], dtype=torch.uint8, device='cuda')
            except Exception:
                continue
            t.fill_(args.poison_byte)
            poison.append(t)
        torch.cuda.synchronize()
        health('before poison steps')
        pl = []
        for i in range(3):
            l = tr.step()
            pl.append(l.clone())
            health(f'poison step {i}', l)
        torch.cuda.synchronize()
        rep['poison_step_losses'] = [float(t.item()) for t in pl]
        hit = []
        recs = sorted(set((sp, nb, f1, f2) for sp, nb, f1, f2, _ in rec))
        for t in poison:
            bad = int((t != args.poison_byte).sum())
            if bad:
                lo, hi = t.data_ptr(), t.data_ptr() + t.numel()
                who = sorted(set(f'{f1}/{f2}' for sp, nb, f1, f2 in recs if sp < hi and sp + nb > lo))
                hit.append(dict(addr=hex(lo), size=t.numel(), changed_bytes=bad, recorded_by=who[:8]))
        rep['poison_tensors'] = len(poison)
        rep['poison_mb'] = round(sum(t.numel() for t in poison) / 1e6, 1)
        rep['poison_hit'] = hit
        del poison

    # the loader loop, bench.loader_leg verbatim + device-side logging (no host synchronisation inside the loop)
    log_sum, log_loss, log_which = [], [], []
    health('before loop')
    if not args.no_prefetch:
        tr.prefetch_batch(*batches[1])
    for i in range(args.steps):
        which = (i + 1) % 3
        if args.no_prefetch:
            tr.set_batch(*batches[which])
        else:
            tr.commit_prefetched()
        log_sum.append({k: checksum(v) for k, v in leaves({k: v for k, v in tr.static.items() if k != 'dloss'})})
        if not args.no_prefetch:
            tr.prefetch_batch(*batches[(i + 2) % 3])
        loss = tr.step()
        log_loss.append(loss.clone())
        log_which.append(which)
        if i < 3:
            health(f'loop step {i}', loss)
    torch.cuda.synchronize()
    losses = [float(t.item()) for t in log_loss]
    wrong = []
    for i, (s, w) in enumerate(zip(log_sum, log_which)):
        for k, v in s.items():
            if k in expect[w] and int(v) != expect[w][k]:
                wrong.append(dict(step=i, leaf=k, got=int(v), want=expect[w][k], matches_other=[j for j in range(3) if expect[j][k] == int(v)]))
    rep['prefetch_losses'] = [round(v, 4) for v in losses]
    rep['static_mismatches'] = wrong[:40]
    rep['n_static_mismatches'] = len(wrong)
    rep['health'] = health_log
    del tr
    gc.collect(); torch.cuda.empty_cache()

    # reference trajectory: identically initialised Trainer, set_batch per step (in-order copies on the compute stream)
    tr, _, _ = bench.build_trainer(batch, True, 0, compact_labels=True)
    tr.set_batch(*batches[0]); tr.step()
    ref = []
    for i in range(args.steps):
        tr.set_batch(*batches[(i + 1) % 3])
        ref.append(tr.step().clone())
    torch.cuda.synchronize()
    ref = [float(t.item()) for t in ref]
    rep['set_batch_losses'] = [round(v, 4) for v in ref]
    rep['max_rel_loss_diff'] = max(abs(a - b) / max(abs(b), 1e-9) for a, b in zip(losses, ref))
    s = json.dumps(rep, indent=1)
    print(s)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(s + '\n')


if __name__ == '__main__':
    main()
