"""What a device-side cross-stream edge costs next to hipGraph replays (round 6, after tools/loader_probe.py found 0.45 ms per step for an edge from
the compute stream to a COPY): the one-rank Trainer's three graph segments, with a stand-in for a gradient collective (a 64-MB device-to-device
copy KERNEL on a second stream) launched after segments 0 and 1 the way Trainer._issue() launches all-reduces in a multi-rank step:
    event on the compute stream -> second stream waits for it -> kernel there -> compute stream waits for the second stream before the last segment.
Arms: plain step | side kernels without any edge | edges without side kernels | both (the multi-rank pattern) | host-side wait instead of the
device-side edge.      python tools/stream_edge_probe.py [steps]"""
import sys
import time

sys.path.insert(0, '.')
import torch

import bench
from prismer_amd import ops


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
    one, _, _ = bench.build_trainer(32, True, 0)            # the Trainer as shipped: the whole one-rank step is ONE graph
    one.step()
    print(f'whole step as ONE graph (shipped)                      {timed(one.step, steps):8.3f} ms/step', flush=True)
    del one
    tr, dims, _ = bench.build_trainer(32, True, 0)
    three_segments(tr)                                      # the probe needs replay boundaries inside the step: forward + decoder backward | encoder backward | optimizer tail
    tr.step()
    side = torch.cuda.Stream()
    a = torch.empty(16 << 20, dtype=torch.float32, device='cuda')
    b = torch.empty_like(a)
    graphs = list(tr.graphs)
    assert len(graphs) == 3 and all(c is None for _, c in graphs)

    def install(edge, kernel, host_wait=False, join=True):
        def after(i):
            def f():
                main_s = torch.cuda.current_stream()
                if edge or host_wait:
                    ev = torch.cuda.Event()
                    ev.record(main_s)
                    if host_wait:
                        ev.synchronize()
                    else:
                        side.wait_event(ev)
                if kernel:
                    with torch.cuda.stream(side):
                        ops.copy_flat(b, a)
                if i == 1 and join and (edge or host_wait):
                    main_s.wait_stream(side)
            return f
        tr.graphs = [(graphs[0][0], after(0)), (graphs[1][0], after(1)), graphs[2]]

    print(f'plain step, three graphs                               {timed(tr.step, steps):8.3f} ms/step', flush=True)
    for name, kw in (('side kernels, no edges', dict(edge=False, kernel=True)),
                     ('edges (event -> side stream waits -> join), no kernels', dict(edge=True, kernel=False)),
                     ('edges + side kernels (the multi-rank pattern)', dict(edge=True, kernel=True)),
                     ('edges + side kernels, no join before the last segment', dict(edge=True, kernel=True, join=False)),
                     ('host waits for the segment, then side kernels, join', dict(edge=False, kernel=True, host_wait=True))):
        install(**kw)
        print(f'{name:54s} {timed(tr.step, steps):8.3f} ms/step', flush=True)
    # how the cost scales with the NUMBER of edges: n extra (event -> wait on its own stream) pairs after segment 0, nothing joined
    sides = [torch.cuda.Stream() for _ in range(8)]
    for n in (1, 2, 4, 8):
        def after0(n=n):
            main_s = torch.cuda.current_stream()
            for j in range(n):
                ev = torch.cuda.Event()
                ev.record(main_s)
                sides[j].wait_event(ev)
                with torch.cuda.stream(sides[j]):
                    ops.copy_flat(b[j << 20:(j + 1) << 20], a[j << 20:(j + 1) << 20])
        tr.graphs = [(graphs[0][0], after0), graphs[1], graphs[2]]
        print(f'{n} edge(s) after segment 0, one small kernel behind each    {timed(tr.step, steps):8.3f} ms/step', flush=True)
    # ONE event, n waiters
    def after0_shared():
        main_s = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main_s)
        for j in range(4):
            sides[j].wait_event(ev)
            with torch.cuda.stream(sides[j]):
                ops.copy_flat(b[j << 20:(j + 1) << 20], a[j << 20:(j + 1) << 20])
    tr.graphs = [(graphs[0][0], after0_shared), graphs[1], graphs[2]]
    print(f'one event after segment 0, four waiting streams            {timed(tr.step, steps):8.3f} ms/step', flush=True)
    # the edge placed after the LAST segment (nothing of the step follows it on the compute stream)
    def after2():
        main_s = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main_s)
        sides[0].wait_event(ev)
        with torch.cuda.stream(sides[0]):
            ops.copy_flat(b[:1 << 20], a[:1 << 20])
    tr.graphs = [graphs[0], graphs[1], (graphs[2][0], after2)]
    print(f'one edge after the last segment                            {timed(tr.step, steps):8.3f} ms/step', flush=True)
    # the same single edge with raw HIP events of different release scopes (torch.cuda.Event() = hipEventDisableTiming only)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    for name, flags in (('hipEventDisableTiming (torch default)', 0x2), ('| hipEventReleaseToDevice', 0x2 | 0x40000000),
                        ('| hipEventDisableSystemFence', 0x2 | 0x20000000)):
        ev = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(ev), ctypes.c_uint(flags)) == 0

        def after0_raw(ev=ev):
            main_h = torch.cuda.current_stream().cuda_stream
            assert hip.hipEventRecord(ev, ctypes.c_void_p(main_h)) == 0
            assert hip.hipStreamWaitEvent(ctypes.c_void_p(sides[0].cuda_stream), ev, ctypes.c_uint(0)) == 0
            with torch.cuda.stream(sides[0]):
                ops.copy_flat(b[:1 << 20], a[:1 << 20])
        tr.graphs = [(graphs[0][0], after0_raw), graphs[1], graphs[2]]
        print(f'one edge, raw event {name:34s} {timed(tr.step, steps):8.3f} ms/step', flush=True)
    tr.graphs = graphs
    print(f'plain step again                                       {timed(tr.step, steps):8.3f} ms/step', flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.copy_flat(b, a)
    torch.cuda.synchronize()
    print(f'the 64-MB copy kernel alone                            {(time.perf_counter() - t0) / 20 * 1e3:8.3f} ms', flush=True)


if __name__ == '__main__':
    main()
