"""Where does a block of attn_bwd_dq_res_kernel spend its time?  (-DPH_TIMELINE build of attention.hip: tools/build_variant.py tl attention.hip:-DPH_TIMELINE;
PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_tl.so python tools/attn_timeline.py)   stamps: 0 entry | 1 staging + Q/dO requests issued | 2 K / V visible (barrier) |
3 first sub-tile: P / dP pass done | 4 delta known | 5 dS -> dQ done | 6 first sub-tile stored | 8 all sub-tiles done | 9 stores acknowledged"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from prismer_amd import _lib, ops

lib = _lib.lib
lib.ph_tl_fetch_attn.restype = C.c_int
lib.ph_tl_fetch_attn.argtypes = [C.c_void_p, C.c_int, C.c_int]
SLOTS, BLOCKS = 16, 4096
BF = torch.bfloat16
for B, H in [tuple(int(v) for v in p.split('x')) for p in os.environ.get('ATTN_TL_CASES', '32x12,21x12').split(',')]:
    S, dh = 260, 64
    W = H * dh
    q = torch.randn(B * S, W, device='cuda').to(BF); kv = torch.randn(B * S, 2 * W, device='cuda').to(BF); do = torch.randn(B * S, W, device='cuda').to(BF)
    ks, qs = (S * 2 * W, 2 * W), (S * W, W)
    o, lse = ops.attention_fwd(q, kv[:, :W], kv[:, W:], B, H, S, S, dh, q_strides=qs, k_strides=ks, v_strides=ks)
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    for rep in range(3):
        buf = np.zeros(BLOCKS * 2 * SLOTS, dtype=np.uint64)
        lib.ph_tl_fetch_attn(buf.ctypes.data, buf.size, 1)
        ops.attention_bwd(do, q, kv[:, :W], kv[:, W:], o, lse, B, H, S, S, dh, q_strides=qs, k_strides=ks, v_strides=ks, dq=dq, dk=dkv[:, :W], dv=dkv[:, W:],
                          dq_strides=qs, dk_strides=ks, dv_strides=ks)
        torch.cuda.synchronize()
    lib.ph_tl_fetch_attn(buf.ctypes.data, buf.size, 1)
    t = buf.reshape(BLOCKS, 2, SLOTS)[:B * H, 0].astype(np.float64)
    rate = np.median((t[:, 9] - t[:, 0])) / max(np.median((t[:, 11] - t[:, 10])) / 100.0, 1e-9)       # ticks per us (s_memrealtime: 100 MHz)
    seg = [('entry->issued', 0, 1), ('->staged', 1, 2), ('P/dP pass', 2, 3), ('delta', 3, 4), ('dS->dQ', 4, 5), ('store+next', 5, 6), ('rest of sub-tiles', 6, 8), ('ack', 8, 9), ('TOTAL', 0, 9)]
    print(f'heads {B * H}: {rate:.0f} ticks/us; block start spread {(t[:, 0].max() - t[:, 0].min()) / rate:.1f} us, end spread {(t[:, 9].max() - t[:, 9].min()) / rate:.1f} us')
    print('   ' + ' | '.join(f'{n} {np.median(t[:, b] - t[:, a]) / rate:5.2f}/{np.percentile(t[:, b] - t[:, a], 90) / rate:5.2f}' for n, a, b in seg) + '   (median / p90 us, wave 0 of each block)')
