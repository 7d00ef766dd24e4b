"""CPU timing of the REFERENCE module classes (build container only; /root/reference is not on the GPU box): Prismer-BASE caption
fine-tune step (forward + backward + torch.optim.AdamW, freeze_vision, fp32, dropout 0.1, train-mode BatchNorm) on the synthetic
inputs of prismer_amd/synth.py.  The number BASELINE.md quotes as "reference CPU path" (SURVEY 8d); bench.py's cpu_baseline leg
times the oracle port on the GPU box's host instead.
    python tests/tools/ref_cpu_timing.py [batch] [steps]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_harness as RH
from prismer_amd import config, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.set_num_threads(os.cpu_count())
d = config.prismer_base()
esd, dsd = synth.synth_encoder_state(d, 0), synth.synth_decoder_state(d, 0)
enc, dec = RH.build_reference(d, esd, dsd)
holder = RH.reference_freeze(enc, dec, 'freeze_vision')
enc.train(); dec.train()
opt = torch.optim.AdamW([p for p in holder.parameters() if p.requires_grad], lr=5e-5, weight_decay=0.05)
x = synth.synth_experts(d, B, seed=1)
ids, mask, labels = synth.synth_text(d, B, 30, seed=1)
times = []
for i in range(steps + 1):
    t0 = time.time()
    random.seed(i)
    e = enc(x)
    out = dec(ids, attention_mask=mask, encoder_hidden_states=e.permute(1, 0, 2), labels=labels, return_dict=True)
    loss = out.loss.mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
    times.append(time.time() - t0)
    print(f'step {i}: {times[-1]:.2f} s  loss {loss.item():.3f}', flush=True)
best = min(times[1:])
print(f'reference classes, Prismer-BASE caption train step, batch {B}, fp32, {os.cpu_count()} threads: best {best:.2f} s/step = {B / best:.2f} images/s')
