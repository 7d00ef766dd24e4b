"""Where the loader-fed step loses time against the HBM-resident one (bench.py `secondary.loader` vs the headline): the same compact-label
Trainer timed (A) on a static batch, (B) with the device-to-device commit of a staged batch every step but no host copy, (C) with the host
copy on the copy stream but nothing waiting for it, (D) the full loader loop with prefetch_batch BEFORE step, (E) with prefetch_batch AFTER step
(the recommended order), then hand-made variants of the two cross-stream edges (ready: compute stream waits for the copy; free: the copy waits
for the previous commit) -- the finding of round 6: a device-side `free` edge costs the replayed step 0.45 ms, a host-side wait nothing.
    python tools/loader_probe.py [steps]"""
import sys
import time

sys.path.insert(0, '.')
import torch

import bench


def three_segments(tr):
    """the one-rank schedule of rounds 1-6 (three replays per step), installed before the first step captures"""
    nl = len(tr.dec_prog.layers)
    tr._schedule = lambda: [(lambda: tr._seg_forward(tr.static, (nl, 0)), None), (tr._seg_enc_backward_with_dec_adamw, None), (tr._seg_optimizer_tail, None)]


def timed(fn, steps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    tr, dims, _ = bench.build_trainer(32, True, 0, compact_labels=True)
    three_segments(tr)                             # (one variant below releases the copy after the first of three replays; the shipped one-rank step is ONE graph, 0.18 ms faster)

    def pin(t):
        return {k: pin(v) for k, v in t.items()} if isinstance(t, dict) else t.cpu().pin_memory()
    batches = []
    for i in range(3):
        x, ids, mask, labels = bench.make_inputs(dims, 32, 30, 4321 + i, torch.device('cuda'), True)
        batches.append((pin(x), ids.cpu().pin_memory(), mask.cpu().pin_memory(), labels.cpu().pin_memory()))
    tr.set_batch(*batches[0])
    tr.step()
    a = timed(tr.step, steps)
    print(f'A static batch                       {a:8.3f} ms/step', flush=True)
    tr.prefetch_batch(*batches[1]); tr.commit_prefetched()            # (allocates the staging set)

    def b():
        tr._staged = True
        tr.commit_prefetched()
        tr.step()
    print(f'B + device-to-device commit          {timed(b, steps):8.3f} ms/step', flush=True)
    k = [0]

    def c():
        k[0] += 1
        with torch.cuda.stream(tr.copy_stream):
            tr._bind(tr.staging, *batches[k[0] % 3], None)
        tr.step()
    print(f'C + host copy on the copy stream     {timed(c, steps):8.3f} ms/step   (nothing waits for it)', flush=True)
    torch.cuda.synchronize()
    tr._staged = False
    tr.prefetch_batch(*batches[1])

    def d():
        k[0] += 1
        tr.commit_prefetched()
        tr.prefetch_batch(*batches[k[0] % 3])
        tr.step()
    print(f'D loader loop, prefetch before step  {timed(d, steps):8.3f} ms/step', flush=True)

    def e():                                       # the host copy enqueued AFTER the step's replays instead of before them
        k[0] += 1
        tr.commit_prefetched()
        tr.step()
        tr.prefetch_batch(*batches[k[0] % 3])
    print(f'E loader loop, prefetch after step   {timed(e, steps):8.3f} ms/step', flush=True)
    def moves():
        def move(dst, src):
            if isinstance(dst, dict):
                for kk in dst:
                    move(dst[kk], src[kk])
            elif dst is not None:
                dst.copy_(src, non_blocking=True)
        move(tr.static, tr.staging)

    def variant(wait_ready, wait_free, host_sync):
        def f():
            k[0] += 1
            cur = torch.cuda.current_stream()
            if host_sync:
                tr._staging_ready.synchronize()
            elif wait_ready:
                cur.wait_event(tr._staging_ready)
            moves()
            tr._staging_free.record(cur)
            if wait_free:
                tr.copy_stream.wait_event(tr._staging_free)
            with torch.cuda.stream(tr.copy_stream):
                tr._bind(tr.staging, *batches[k[0] % 3], None)
                tr._staging_ready.record(tr.copy_stream)
            tr.step()
        return f
    tr.commit_prefetched()
    for name, args in (('both edges (= D)', (True, True, False)), ('no ready edge', (False, True, False)), ('no free edge', (True, False, False)),
                       ('no edges', (False, False, False)), ('host waits for the copy, free edge', (True, True, True))):
        print(f'  {name:40s} {timed(variant(*args), steps):8.3f} ms/step', flush=True)
    # the host copy as a KERNEL that reads the pinned pages (no copy engine), both edges kept
    from prismer_amd import ops

    def kcopy(dst, src):
        if isinstance(dst, dict):
            for kk in dst:
                kcopy(dst[kk], src[kk])
        elif dst is not None and dst.numel() * dst.element_size() >= 4096 and dst.shape == src.shape:
            ops.copy_flat(dst.view(-1), src.view(-1))
        elif dst is not None:
            dst[:, :src.shape[1]].copy_(src, non_blocking=True) if dst.dim() == 2 and src.shape != dst.shape else dst.copy_(src, non_blocking=True)

    def kvariant():
        k[0] += 1
        cur = torch.cuda.current_stream()
        cur.wait_event(tr._staging_ready)
        moves()
        tr._staging_free.record(cur)
        tr.copy_stream.wait_event(tr._staging_free)
        with torch.cuda.stream(tr.copy_stream):
            b = batches[k[0] % 3]
            kcopy(tr.staging['experts'], b[0])
            for key, t in zip(('input_ids', 'attention_mask', 'labels'), b[1:]):
                tr.staging[key][:, :t.shape[1]].copy_(t, non_blocking=True)
            tr._staging_ready.record(tr.copy_stream)
        tr.step()
    try:
        print(f'  {"both edges, copy by a kernel reading pinned pages":40s} {timed(kvariant, steps):8.3f} ms/step', flush=True)
    except Exception as e:
        print('  kernel-copy variant failed:', type(e).__name__, e, flush=True)
    # both edges, but the staging set is released (and the next host copy starts) only after the step's FIRST graph segment
    mid = torch.cuda.Event()
    g0, c0 = tr.graphs[0]
    tr.graphs[0] = (g0, lambda: mid.record(torch.cuda.current_stream()))

    def late():
        k[0] += 1
        cur = torch.cuda.current_stream()
        cur.wait_event(tr._staging_ready)
        moves()
        tr.step()                                  # records `mid` after segment 0
        tr.copy_stream.wait_event(mid)
        with torch.cuda.stream(tr.copy_stream):
            tr._bind(tr.staging, *batches[k[0] % 3], None)
            tr._staging_ready.record(tr.copy_stream)
    print(f'  {"both edges, host copy released after segment 0":40s} {timed(late, steps):8.3f} ms/step', flush=True)
    tr.graphs[0] = (g0, c0)

    def hostfree(order_e):
        def f():
            k[0] += 1
            cur = torch.cuda.current_stream()
            cur.wait_event(tr._staging_ready)
            moves()
            tr._staging_free.record(cur)
            if order_e:
                tr.step()
            tr._staging_free.synchronize()             # HOST waits until the commit has read the staging set; the copy carries no device-side edge
            with torch.cuda.stream(tr.copy_stream):
                tr._bind(tr.staging, *batches[k[0] % 3], None)
                tr._staging_ready.record(tr.copy_stream)
            if not order_e:
                tr.step()
        return f
    print(f'  {"host-side free wait, prefetch after step":40s} {timed(hostfree(True), steps):8.3f} ms/step', flush=True)
    print(f'  {"host-side free wait, prefetch before step":40s} {timed(hostfree(False), steps):8.3f} ms/step', flush=True)
    torch.cuda.synchronize()
    tr._staged = False
    tr.prefetch_batch(*batches[1])
    for rep in range(2):                           # run-to-run spread inside one process
        tr.commit_prefetched()
        print(f'  repeat {rep}: A {timed(tr.step, steps):8.3f}', end='', flush=True)
        tr.prefetch_batch(*batches[1])
        print(f'   D {timed(d, steps):8.3f}   E {timed(e, steps):8.3f}', flush=True)
    # the host copy alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        with torch.cuda.stream(tr.copy_stream):
            tr._bind(tr.staging, *batches[i % 3], None)
    torch.cuda.synchronize()
    print(f'host -> device copy of one batch alone {(time.perf_counter() - t0) * 100:8.3f} ms', flush=True)
    # host time of the three calls (no device wait)
    tr._staged = True
    t0 = time.perf_counter()
    for i in range(10):
        tr.commit_prefetched(); tr.prefetch_batch(*batches[i % 3])
    h = (time.perf_counter() - t0) * 100
    torch.cuda.synchronize()
    print(f'host time of commit_prefetched + prefetch_batch {h:8.3f} ms', flush=True)


if __name__ == '__main__':
    main()
