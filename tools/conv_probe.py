"""The stems' grouped implicit-GEMM convolutions (forward and data gradient, layers 2..4 of the six expert stems at batch 32,
vit.py:88-120) on the register-staged gather kernel (tuning mode 1) and on the 256x128 LDS-DMA kernel with the gather in the DMA
source address (mode 6, round 4).  hipGraph replay, interleaved.
    python tools/conv_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import _lib, ops

BF = torch.bfloat16
B = int(os.environ.get('CONV_B', '32'))


def graph_of(fn, n):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    return g


def layer_items(layer):
    """(fwd items, dgrad items, GFLOP fwd) of stem layer `layer` (1..3 = the convs 96->192, 192->384, 384->768) for 3 dense + 3 label experts"""
    cin, cout = 96 << (layer - 1), 192 << (layer - 1)
    fwd, dg, fl = [], [], 0.0
    for label in (False, True):
        strides = (2, 2, 1, 1) if label else (2, 2, 2, 2)
        H = 56 if label else 224
        for i in range(layer):
            H = ops.conv_out_size(H, 3, strides[i])
        s_ = strides[layer]
        Ho = ops.conv_out_size(H, 3, s_)
        for e in range(3):
            x = (torch.randn(B, H, H, cin, device='cuda') * 0.5).to(BF)
            w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
            Kp = 9 * cin
            shadow = torch.zeros(cout, Kp, dtype=BF, device='cuda')
            ops.conv_weight_to_shadow(w, shadow, cout, cin, 3, Kp)
            y = torch.empty(B * Ho * Ho, cout, dtype=BF, device='cuda')
            slabs = torch.zeros(8, 2, cout, device='cuda', dtype=torch.float64)
            fwd.append((x, (B, H, H, cin, 3, s_), shadow, y, slabs))
            wd = torch.zeros(cin, 9 * cout, dtype=BF, device='cuda')
            ops.conv_dgrad_shadows([(w, wd, cout, cin, s_)])
            dy = (torch.randn(B * Ho * Ho, cout, device='cuda') * 0.3).to(BF)
            dx = torch.empty(B * H * H, cin, dtype=BF, device='cuda')
            dg.append((dy, wd, dx, (B, H, H, cin, cout, s_)))
            fl += 2.0 * B * Ho * Ho * cout * 9 * cin
    return fwd, dg, fl


def single():
    """one problem, no grouping: the implicit gather against the same product on the materialised im2col matrix (plain 256x128 kernel)"""
    for (H, cin, cout, s_) in ((28, 384, 768, 2), (56, 192, 384, 2), (14, 384, 768, 1)):
        Ho = ops.conv_out_size(H, 3, s_)
        x = (torch.randn(B, H, H, cin, device='cuda') * 0.5).to(BF)
        w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
        Kp = 9 * cin
        shadow = torch.zeros(cout, Kp, dtype=BF, device='cuda')
        ops.conv_weight_to_shadow(w, shadow, cout, cin, 3, Kp)
        y = torch.empty(B * Ho * Ho, cout, dtype=BF, device='cuda')
        y2 = torch.empty_like(y)
        col = ops.im2col(x, B, H, H, cin, 3, s_, Kp)
        _lib.lib.ph_gemm_tuning(6, 1)
        fns = [('implicit', lambda: ops.conv_fwd_grouped([(x, (B, H, H, cin, 3, s_), shadow, y, None)])), ('explicit col', lambda: ops.gemm(col, shadow, out=y2))]
        graphs = [graph_of(f, 10) for _, f in fns]
        best = [1e30, 1e30]
        for _ in range(5):
            for i, g in enumerate(graphs):
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(); g.replay(); t1.record(); torch.cuda.synchronize()
                best[i] = min(best[i], t0.elapsed_time(t1) / 10 * 1e3)
        fl = 2.0 * B * Ho * Ho * cout * Kp
        print(f'single conv {cin}->{cout} H{H} s{s_} (M={B * Ho * Ho}): ' + ' | '.join(f'{n}: {best[i]:7.1f} us {fl / best[i] / 1e6:5.0f} TF' for i, (n, _) in enumerate(fns)) +
              f' | equal {torch.equal(y, y2)}', flush=True)


def main():
    torch.manual_seed(0)
    single()
    modes = [1, 6]
    for layer in (1, 2, 3):
        fwd, dg, fl = layer_items(layer)
        for what, fn in (('fwd', lambda: ops.conv_fwd_grouped(fwd)), ('dgrad', lambda: ops.conv_dgrad_grouped(dg))):
            graphs, outs = [], []
            for m in modes:
                _lib.lib.ph_gemm_tuning(m, 128)
                fn(); torch.cuda.synchronize()
                outs.append([it[3].clone() for it in fwd] if what == 'fwd' else [it[2].clone() for it in dg])
                graphs.append(graph_of(fn, 5))
            err = max(((a.float() - b.float()).norm() / b.float().norm()).item() for a, b in zip(outs[0], outs[1]))
            best = [1e30] * len(modes)
            for _ in range(5):
                for i, g in enumerate(graphs):
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record(); g.replay(); t1.record(); torch.cuda.synchronize()
                    best[i] = min(best[i], t0.elapsed_time(t1) / 5 * 1e3)
            print(f'stem conv {layer + 1} ({96 << (layer - 1)}->{192 << (layer - 1)}) {what:5s} 6 experts bs{B}: ' +
                  ' | '.join(f'm{m}: {best[i]:7.1f} us {fl / best[i] / 1e6:5.0f} TF' for i, m in enumerate(modes)) + f' | rel diff {err:.1e}', flush=True)
    _lib.lib.ph_gemm_tuning(*_lib.GEMM_BIG_DEFAULT)


if __name__ == '__main__':
    main()
