"""AdamW kernel bandwidth: product build vs variants (AB_LIBS), 242 M parameters like Prismer-BASE's trainable set."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import _lib, ops

def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGS.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    return lib

libs = [(os.path.basename(p), load(p)) for p in os.environ.get('AB_LIBS', _lib.LIB_PATH).split(',')]
n = 174_000_000
p = torch.randn(n, device='cuda'); g = torch.randn(n, device='cuda') * 1e-3; m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
pb = torch.empty(n, dtype=torch.bfloat16, device='cuda'); hyper = torch.tensor([1e-4, 0.1, 0.001], device='cuda')
for rounds in range(3):
    for name, lib in libs:
        ops.lib = lib
        for zero in (1,):
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5):
                ops.adamw(p, g, m, v, pb, n, hyper, zero_grad=bool(zero))
            t1.record(); torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 5
            print(f'{name:28s} zero_grad={zero}: {ms:.3f} ms  {n * 30 / ms / 1e9:.2f} TB/s', flush=True)
