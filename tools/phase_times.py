"""Where does a training step's wall time go?  Captures each phase of the native step as its own hipGraph and times the
replays with HIP events (steady state, no host launch overhead):

    front fwd (patch embed + expert stems) | trunk fwd (resampler + ViT) | decoder fwd (+CE) | decoder bwd |
    trunk bwd | front bwd | AdamW

    python tools/phase_times.py [batch]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from prismer_amd import ops

F32, BF16 = torch.float32, torch.bfloat16


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    tr, dims, _ = bench.build_trainer(batch, False, 0)
    for _ in range(2):                      # eager warm-up (allocations, shadows)
        tr.step()
    torch.cuda.synchronize()
    s = tr.static
    st = {}
    ep, dp = tr.enc_prog, tr.dec_prog
    B = batch

    def front_f():
        for store in tr.stores:
            store.grad.zero_()
        st['h'], st['xf'], st['svf'] = ep.forward_front(s['experts'], tr.table, True, True)

    def trunk_f():
        st['enc'], st['svt'] = ep.forward_trunk(st['h'], st['xf'], B, True)

    def dec_f():
        st['logits'], st['loss'], st['svd'] = dp.forward(s['input_ids'], s['attention_mask'], st['enc'], s['labels'], tr.seed, True)

    def dec_b():
        dloss = torch.full((B,), 1.0 / B, dtype=F32, device='cuda')
        st['denc'] = dp.backward(st['svd'], dloss)
        ops.join_side()

    def trunk_b():
        d = ep.d
        st['dh'] = torch.empty(B * d.seq_len, d.width, dtype=BF16, device='cuda')
        st['dxf'] = torch.empty(B * d.num_expert_tokens, d.width, dtype=BF16, device='cuda')
        ep.backward_trunk(st['svt'], st['denc'], st['dh'], st['dxf'])
        ops.join_side()

    def front_b():
        ep.backward_front(st['svf'], st['dh'], st['dxf'])
        ops.join_side()

    def opt():
        tr._adamw(1); tr._adamw(0)

    phases = [('front fwd', front_f), ('trunk fwd', trunk_f), ('decoder fwd', dec_f), ('decoder bwd', dec_b),
              ('trunk bwd', trunk_b), ('front bwd', front_b), ('adamw', opt)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _, fn in phases:
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    pool = torch.cuda.graph_pool_handle()
    graphs = []
    for name, fn in phases:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool):
            fn()
        graphs.append((name, g))
    for _ in range(3):
        for _, g in graphs:
            g.replay()
    torch.cuda.synchronize()
    reps = 10
    acc = {n: 0.0 for n, _ in graphs}
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(graphs) + 1)]
        evs[0].record()
        for i, (_, g) in enumerate(graphs):
            g.replay()
            evs[i + 1].record()
        torch.cuda.synchronize()
        for i, (n, _) in enumerate(graphs):
            acc[n] += evs[i].elapsed_time(evs[i + 1])
    t1.record(); torch.cuda.synchronize()
    tot = sum(acc.values()) / reps
    for n, _ in graphs:
        print(f'{n:14s} {acc[n] / reps:7.3f} ms  {100 * acc[n] / reps / tot:5.1f} %')
    print(f'{"sum":14s} {tot:7.3f} ms   (batch {batch}: {batch / tot * 1e3:.0f} img/s if phases ran back to back)')


if __name__ == '__main__':
    main()
