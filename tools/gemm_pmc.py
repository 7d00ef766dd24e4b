"""eager loop of the step's big GEMM shapes (256x128 LDS-DMA kernel) and one grouped conv for rocprofv3 --pmc passes
    rocprofv3 --kernel-trace --pmc <counters> -d out -o pmc -- python tools/gemm_pmc.py;  python tools/pmc_dump.py out/.../pmc_results.db gemm"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import _lib, ops
BF = torch.bfloat16
torch.manual_seed(0)
_lib.lib.ph_gemm_tuning(6, 1)
a = (torch.randn(8320, 3072, device='cuda') * 0.5).to(BF)
w = (torch.randn(768, 3072, device='cuda') * 0.05).to(BF)
a2 = (torch.randn(8320, 768, device='cuda') * 0.5).to(BF)
w2 = (torch.randn(768, 768, device='cuda') * 0.05).to(BF)
wt = (torch.randn(3072, 768, device='cuda') * 0.05).to(BF)          # [K, N] operand of a data gradient
o = torch.empty(8320, 768, dtype=BF, device='cuda')
for _ in range(6):
    ops.gemm(a, w, out=o)                       # K = 3072: the k loop dominates
    ops.gemm(a2, w2, out=o)                     # K = 768
    ops.gemm(a, wt, out=o, trans_b=True)        # [K,N] B operand (ds_read_b64_tr_b16 fragments)
torch.cuda.synchronize()
_lib.lib.ph_gemm_tuning(*_lib.GEMM_BIG_DEFAULT)
