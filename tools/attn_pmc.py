"""eager loop of the ViT-shaped attention kernels for rocprofv3 --pmc passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import ops
BF = torch.bfloat16
B, H, S, dh = 32, 12, 260, 64
W = H * dh
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * W, device='cuda').to(BF)
do = torch.randn(B * S, W, device='cuda').to(BF)
st = (S * 3 * W, 3 * W)
dqkv = torch.empty_like(qkv)
for _ in range(12):
    o, lse = ops.attention_fwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], B, H, S, S, dh, q_strides=st, k_strides=st, v_strides=st)
    ops.attention_bwd(do, qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], o, lse, B, H, S, S, dh, q_strides=st, k_strides=st, v_strides=st,
                      dq=dqkv[:, :W], dk=dqkv[:, W:2 * W], dv=dqkv[:, 2 * W:], dq_strides=st, dk_strides=st, dv_strides=st)
torch.cuda.synchronize()
