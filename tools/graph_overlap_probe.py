"""Do two branch-free hipGraphs launched on two streams overlap on this runtime?  Probe for micro-batch pipelining: the decoder's
latency-bound chain (forward + backward, own buffers) as graph D, the trunk's MFMA-bound forward as graph T.
    python tools/graph_overlap_probe.py [batch]"""
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
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    s = tr.static
    ep, dp = tr.enc_prog, tr.dec_prog
    B = batch
    st = {}
    h, xf, _ = ep.forward_front(s['experts'], tr.table, True, False)
    enc_out, _ = ep.forward_trunk(h, xf, B, False)
    enc_static = enc_out.clone()
    torch.cuda.synchronize()

    def trunk_f():
        st['enc'], st['svt'] = ep.forward_trunk(h, xf, B, True)

    def dec_fb():
        _, loss, sv = dp.forward(s['input_ids'], s['attention_mask'], enc_static, s['labels'], tr.seed, True)
        dloss = torch.full((B,), 1.0 / B, dtype=F32, device='cuda')
        st['denc'] = dp.backward(sv, dloss)
        ops.join_side()

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    graphs = {}
    for name, fn, strm in (('trunk_fwd', trunk_f, s1), ('decoder_fwd_bwd', dec_fb, s2)):
        strm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(strm):
            fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=strm):
            fn()
        graphs[name] = (g, strm)
    torch.cuda.synchronize()

    def timed(names, reps=10):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(reps):
            evs = []
            for n in names:
                g, strm = graphs[n]
                strm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(strm):
                    g.replay()
                evs.append(strm)
            for strm in evs:
                torch.cuda.current_stream().wait_stream(strm)
        t1.record(); torch.cuda.synchronize()
        return t0.elapsed_time(t1) / reps
    a = timed(['trunk_fwd']); b = timed(['decoder_fwd_bwd']); c = timed(['trunk_fwd', 'decoder_fwd_bwd'])
    print(f'batch {batch}: trunk fwd alone {a:.2f} ms | decoder fwd+bwd alone {b:.2f} ms | both, two streams {c:.2f} ms  '
          f'(serial = {a + b:.2f}, perfect overlap = {max(a, b):.2f})')


if __name__ == '__main__':
    main()
