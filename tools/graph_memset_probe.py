"""Is a memset node of a captured single-stream hipGraph executed IN ORDER with the kernel nodes around it on this ROCm?

Round-5 finding behind the garbage `loader` leg of round 4 (profiles/r5_graph_memset_order.txt): `ph_ce_fwd` zeroed its per-sample loss
accumulator with hipMemsetAsync; captured, that is a MEMSET node between kernel nodes.  This probe captures, on one stream,

    kernel  x[:] = 7        (a kernel node)
    memset  x[:] = 0        (hipMemsetAsync -> memset node)
    kernel  x[:] += 1       (a kernel node)

and replays it: correct execution leaves 1.0 in every element on every replay (measured: junk from the second replay on for 128-byte nodes).  Same for torch.zeros / Tensor.zero_()
inside a capture (are they memset nodes or fill kernels?).
"""
import ctypes
import json
import sys

import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def run(n, pad_kernels, how):
    x = torch.full((n,), 3.0, device='cuda')
    junk = torch.zeros(1 << 20, device='cuda')
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(pad_kernels):              # kernel nodes in front (the real graph has hundreds)
            junk.add_(1.0)
        x.fill_(7.0)
        if how == 'hipMemsetAsync':
            rc = hip.hipMemsetAsync(x.data_ptr(), 0, n * 4, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        elif how == 'zero_':
            x.zero_()
        elif how == 'fill_0':
            x.fill_(0.0)
        x.add_(1.0)
        for _ in range(pad_kernels):
            junk.add_(1.0)
    out = []
    for _ in range(5):
        g.replay()
        torch.cuda.synchronize()
        out.append(sorted(set(x.tolist())))
    return out


def main():
    res = {}
    for how in ('hipMemsetAsync', 'zero_', 'fill_0'):
        for n in (32, 1 << 16):
            for pad in (0, 50):
                res[f'{how} n={n} kernels_around={pad}'] = run(n, pad, how)
    verdict = {k: ('CORRECT' if all(v == [1.0] for v in vs) else 'WRONG') for k, vs in res.items()}
    print(json.dumps(dict(torch=torch.__version__, hip=torch.version.hip, device=torch.cuda.get_device_name(0), values=res, verdict=verdict), indent=1))
    return 0


if __name__ == '__main__':
    sys.exit(main())
