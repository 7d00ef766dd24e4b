"""fused-epilogue GEMM probe under hipGraph replay: the ViT / decoder shapes with the epilogues the step really uses."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prismer_amd import ops
from prismer_amd._lib import ACT_QUICKGELU, ACT_GELU
from tools.gemm_probe_small import graph_time
BF = torch.bfloat16


def probe(name, M, N, K, layout='nt', **kw):
    dev = 'cuda'
    if layout == 'nt':
        a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF); lk = {}
    elif layout == 'nn':
        a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(K, N, device=dev).to(BF); lk = dict(trans_b=True)
    else:
        a = torch.randn(K, M, device=dev).to(BF); b = torch.randn(K, N, device=dev).to(BF); lk = dict(trans_a=True, trans_b=True)
    args = {}
    if kw.get('bias'):
        args['bias'] = torch.randn(N, device=dev)
    if kw.get('act'):
        args['act'] = kw['act']
    if kw.get('pre_out'):
        args['pre_out'] = torch.empty(M, N, dtype=BF, device=dev)
    if kw.get('act_in'):
        args['act_in'] = torch.randn(M, N, device=dev).to(BF)
    if kw.get('residual') == 'bf16':
        args['residual'] = torch.randn(M, N, device=dev).to(BF)
    elif kw.get('residual') == 'f32':
        args['residual'] = torch.randn(M, N, device=dev)
    if kw.get('drop'):
        seed = torch.tensor([1234], dtype=torch.int64, device=dev)
        args['drop'] = ops.Dropout(0.1, seed, 7)
    f32 = kw.get('out_f32', False)
    if f32:
        args.update(out_f32=True, accumulate=True)
    out = torch.zeros(M, N, dtype=torch.float32 if f32 else BF, device=dev)
    t = graph_time(lambda: ops.gemm(a, b, out=out, **lk, **args))
    print(f'{name:28s} {layout} M={M:6d} N={N:5d} K={K:5d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.0f} TF', flush=True)


if __name__ == '__main__':
    probe('vit plain', 8320, 3072, 768)
    probe('vit c_fc bias+qgelu+pre', 8320, 3072, 768, bias=True, act=ACT_QUICKGELU, pre_out=True)
    probe('vit dgrad c_proj act_in', 8320, 3072, 768, 'nn', act=ACT_QUICKGELU, act_in=True)
    probe('vit qkv bias', 8320, 2304, 768, bias=True)
    probe('vit out bias+res', 8320, 768, 768, bias=True, residual='bf16')
    probe('vit out plain', 8320, 768, 768)
    probe('vit c_proj bias+res', 8320, 768, 3072, bias=True, residual='bf16')
    probe('vit dgrad qkv', 8320, 768, 2304, 'nn')
    probe('vit wgrad 768', 768, 768, 8320, 'tn', out_f32=True)
    probe('vit wgrad fc', 3072, 768, 8320, 'tn', out_f32=True)
    probe('dec dense bias+drop+resf32', 960, 768, 768, bias=True, drop=True, residual='f32')
    probe('dec dense plain', 960, 768, 768)
    probe('dec up bias+gelu+pre', 960, 3072, 768, bias=True, act=ACT_GELU, pre_out=True)
    probe('dec down bias+drop+resf32', 960, 768, 3072, bias=True, drop=True, residual='f32')
    probe('dec wgrad', 768, 768, 960, 'tn', out_f32=True)
    probe('lm head bias', 960, 50304, 768, bias=True)
    probe('resampler kv', 39680, 1536, 768, bias=True)
