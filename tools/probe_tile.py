import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import ab_probe as P
P.gemm_case('out b+res', 8320, 768, 768, bias=True, residual='bf16')
P.gemm_case('adaptor down', 8320, 384, 768, bias=True, act=P.ACT_QUICKGELU, pre_out=True)
P.gemm_case('adaptor up', 8320, 768, 384, bias=True, residual='bf16')
P.gemm_case('c_proj b+res', 8320, 768, 3072, bias=True, residual='bf16')
P.gemm_case('dgrad out', 8320, 768, 768, 'nn')
P.gemm_case('qkv bias', 8320, 2304, 768, bias=True)
P.gemm_case('resampler q', 2048, 768, 768, bias=True)
P.gemm_case('conv2 fwd', 100352, 192, 864)
P.gemm_case('conv2 dgrad', 100352, 864, 192, 'nn')
P.gemm_case('conv3 fwd dense', 25088, 384, 1728)
