"""Step-level A/B inside ONE process on ONE box: one bench-configuration Trainer per arm (hipGraph-captured under that arm's settings), the
arms timed interleaved (A, B, A, B, ...) so that box / clock drift cannot pass for a change.

    python tools/step_ab.py gemm6 gemm7              # arms: ph_gemm_tuning mode during warm-up + capture (dispatch is frozen into the graphs)
    python tools/step_ab.py base base                # arms defined in ARMS below (trainer=dict(...) passes Trainer keyword arguments)

Prints ms per step per arm (best and median of the rounds) and the loss after the last step (all arms start from the same weights)."""
import os
import sys
import statistics

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from prismer_amd import _lib

ARMS = {
    'base': dict(),
    'r5_kernels': dict(gemm_mode=6, attn_mode=2),   # the round-5 kernel selection: LEAN loop without SPREAD, streaming attention backward
    'gemm6': dict(gemm_mode=6),
    'gemm7': dict(gemm_mode=7),
    'attn_stream': dict(attn_mode=2),            # ph_attention_tuning(2): without the head-resident backward kernels
    'attn_res': dict(attn_mode=1),
    'ln_old': dict(ln_mode=0),                   # ph_layernorm_tuning(0): one-row-per-wave LayerNorm kernels
    'ln_new': dict(ln_mode=1),
}


def make(arm):
    spec = ARMS[arm]
    if 'gemm_mode' in spec:
        _lib.lib.ph_gemm_tuning(spec['gemm_mode'], -1)
    if 'ln_mode' in spec:
        _lib.lib.ph_layernorm_tuning(spec['ln_mode'])
    if 'attn_mode' in spec:
        _lib.lib.ph_attention_tuning(spec['attn_mode'])
    restore = []
    extra = spec.get('trainer', {})
    if extra:
        from prismer_amd import trainer as T
        orig = T.Trainer.__init__

        def patched(self, *a, **k):
            k.update(extra)
            orig(self, *a, **k)
        T.Trainer.__init__ = patched
    try:
        tr, dims, n_train = bench.build_trainer(int(os.environ.get('AB_BATCH', '32')), True, 0, workload=os.environ.get('AB_WORKLOAD', 'base_caption'))
        for _ in range(3):
            loss = tr.step()
        torch.cuda.synchronize()
    finally:
        for r in restore:
            r()
        if extra:
            T.Trainer.__init__ = orig
        _lib.lib.ph_gemm_tuning(*_lib.GEMM_BIG_DEFAULT)
        _lib.lib.ph_attention_tuning(1)
        _lib.lib.ph_layernorm_tuning(1)
    return tr


def main():
    arms = sys.argv[1:] or ['base', 'base']
    trs = [make(a) for a in arms]
    steps, rounds = int(os.environ.get('AB_STEPS', '20')), int(os.environ.get('AB_ROUNDS', '5'))
    times = [[] for _ in arms]
    losses = [None] * len(arms)
    for r in range(rounds):
        for i, tr in enumerate(trs):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                loss = tr.step()
            e1.record(); torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / steps)
            losses[i] = float(loss.item())
    for a, t, l in zip(arms, times, losses):
        print(f'{a:16s} best {min(t):7.3f} ms  median {statistics.median(t):7.3f} ms  rounds {" ".join(f"{x:.3f}" for x in t)}  loss {l:.4f}', flush=True)


if __name__ == '__main__':
    main()
