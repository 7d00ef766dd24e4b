"""minimal hipGraph stream-capture patterns: which cross-stream dependency shapes survive capture_end on this ROCm?"""
import sys, torch
mode = sys.argv[1]
x = torch.zeros(1 << 20, device='cuda')
M, P = torch.cuda.Stream(), torch.cuda.Stream()
def fork(src, dst):
    ev = torch.cuda.Event(); ev.record(src); dst.wait_event(ev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    O = torch.cuda.current_stream()
    x.add_(1)
    if mode == 'flat':            # O -> M, O -> P, joins to O
        fork(O, M); fork(O, P)
        with torch.cuda.stream(M): a = x * 2
        with torch.cuda.stream(P): b = x * 3
        O.wait_stream(M); O.wait_stream(P)
    elif mode == 'nested':        # O -> M -> P ; P -> M -> O
        fork(O, M)
        with torch.cuda.stream(M):
            a = x * 2
            fork(M, P)
            with torch.cuda.stream(P): b = a * 3
            M.wait_stream(P)
            c = a + b
        O.wait_stream(M)
    elif mode == 'nested_join_both':
        fork(O, M)
        with torch.cuda.stream(M):
            a = x * 2
            fork(M, P)
            with torch.cuda.stream(P): b = a * 3
            M.wait_stream(P)
            c = a + b
        O.wait_stream(M); O.wait_stream(P)
    elif mode == 'prefork':       # P enters the capture from O first, later exchanges edges with M
        fork(O, M); fork(O, P)
        with torch.cuda.stream(P): b0 = x * 5
        with torch.cuda.stream(M):
            a = x * 2
            fork(M, P)
            with torch.cuda.stream(P): b = a * 3
            M.wait_stream(P)
            c = a + b
        O.wait_stream(M); O.wait_stream(P)
    elif mode == 'multi':         # repeated edges M <-> P like the side stream
        fork(O, M)
        with torch.cuda.stream(M):
            a = x * 2
            for _ in range(3):
                fork(M, P)
                with torch.cuda.stream(P): b = a * 3
                a = a + 1
            M.wait_stream(P)
            c = a + b
        O.wait_stream(M)
    if mode == 't1':              # P pre-forked from O; edge M -> P; P joins O only (what the product does with the side stream)
        fork(O, M); fork(O, P)
        with torch.cuda.stream(M):
            a = x * 2
            fork(M, P)
            with torch.cuda.stream(P): b = a * 3
            c = a + 1
        O.wait_stream(M); O.wait_stream(P)
    elif mode == 't2':            # P enters the capture through M; joins O only
        fork(O, M)
        with torch.cuda.stream(M):
            a = x * 2
            fork(M, P)
            with torch.cuda.stream(P): b = a * 3
            c = a + 1
        O.wait_stream(M); O.wait_stream(P)
    elif mode == 't3':            # P pre-forked, independent work; M waits P (edge P -> M, no diamond)
        fork(O, M); fork(O, P)
        with torch.cuda.stream(P): b0 = x * 5
        with torch.cuda.stream(M):
            a = x * 2
            M.wait_stream(P)
            c = a + b0
        O.wait_stream(M); O.wait_stream(P)
    elif mode == 't4':            # diamond through events only: M -> P -> M with P pre-forked, join both to O
        fork(O, M); fork(O, P)
        with torch.cuda.stream(M):
            a = x * 2
            fork(M, P)
        with torch.cuda.stream(P):
            b = a * 3
            ev = torch.cuda.Event(); ev.record(P)
        with torch.cuda.stream(M):
            M.wait_event(ev)
            c = a + b
        O.wait_stream(M); O.wait_stream(P)
print(mode, 'capture ok', flush=True)
g.replay(); torch.cuda.synchronize()
print(mode, 'replay ok', float(x[0]), flush=True)
