"""What does ONE dependent launch cost inside a replayed hipGraph chain on this box?  Chains of 40 identical launches on one stream:
an empty-ish kernel (ph_advance_seed: one thread), a 1-block kernel, a small LayerNorm, the decoder's GEMM.  `python tools/launch_floor_probe.py`"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prismer_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dec_gemm_probe import chain_time

BF = torch.bfloat16


def main():
    seed = torch.tensor([77], dtype=torch.int64, device='cuda')
    x = torch.randn(960, 768, device='cuda')
    xb = x.to(BF)
    g, b = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
    out, outf = torch.empty(960, 768, device='cuda', dtype=BF), torch.empty(960, 768, device='cuda')
    w = (torch.randn(768, 768, device='cuda') * 0.05).to(BF)
    i64 = torch.zeros(8, dtype=torch.int64, device='cuda')
    small = torch.randn(64, 768, device='cuda')
    cases = [('ph_advance_seed (1 thread)', lambda: ops.advance_seed(seed)),
             ('ph_add_i64 (8 elements)', lambda: ops.add_i64(i64, 1)),
             ('layernorm_fwd 64 x 768 fp32', lambda: ops.layernorm_fwd(small, g, b, out=out[:64], out_f32=outf[:64])),
             ('layernorm_fwd 960 x 768 fp32 -> bf16 + fp32', lambda: ops.layernorm_fwd(x, g, b, out=out, out_f32=outf)),
             ('gemm 960x768x768 plain', lambda: ops.gemm(xb, w, out=out)),
             ('gemm 64x64x768 (one tile)', lambda: ops.gemm(xb[:64], w[:64], out=out[:64, :64])),
             ('torch add_ 960x768 (elementwise)', lambda: outf.add_(1.0))]
    for name, fn in cases:
        print(f'{name:50s} {chain_time(fn):6.2f} us per launch in a chain of 40')
    # trunk LayerNorm (8320 bf16 rows of 768): one row per wave with 8-B vectors vs half a wave per row with 16-B vectors (round 5)
    xt = torch.randn(8320, 768, device='cuda').to(BF)
    ot = torch.empty_like(xt)
    for flag, what in ((0, 'one row per wave, 8-B vectors'), (1, 'half a wave per row, 16-B vectors')):
        _lib.lib.ph_layernorm_tuning(flag)
        t = chain_time(lambda: ops.layernorm_fwd(xt, g, b, out=ot))
        print(f'layernorm_fwd 8320 x 768 bf16, {what:36s} {t:6.2f} us per launch = {8320 * 768 * 4 / t / 1e6:.2f} TB/s')
    _lib.lib.ph_layernorm_tuning(1)


if __name__ == '__main__':
    main()
