"""Instruction mix of the basic blocks of one kernel in a hipcc -S listing (CPU-only analysis aid).

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only prismer_amd/csrc/attention.hip -o /tmp/attn.s
    python tools/isa_mix.py /tmp/attn.s attn_fwd_kernelILi64ELb1ELi1E

Prints, per basic block with at least `min` instructions, the count of MFMA / VALU / transcendental / LDS / VMEM / SALU
instructions and an issue-cycle estimate for one wave64 (VALU 4, transcendental 16, packed-f32 4, MFMA 16x16x32 16 cycles).
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')):
        return 'trans'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt') or op.startswith('s_barrier') or op.startswith('s_nop'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(':') is False and re.match(r'^_Z\S+:', l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
    blocks, cur = [], ['entry', []]
    for l in lines[start + 1:end]:
        m = re.match(r'^(\.LBB\S+):', l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
            continue
        t = l.strip()
        if not t or t.startswith((';', '.')):
            continue
        cur[1].append(t.split()[0])
    blocks.append(cur)
    tot = Counter()
    for name, ops in blocks:
        c = Counter(classify(o) for o in ops)
        tot.update(c)
        if len(ops) < lo:
            continue
        cyc = 4 * c['valu'] + 16 * c['trans']
        top = Counter(o for o in ops if classify(o) in ('valu', 'trans')).most_common(14)
        print(f"{name:12s} n={len(ops):4d} mfma={c['mfma']:3d} valu={c['valu']:4d} trans={c['trans']:3d} lds={c['lds']:3d} vmem={c['vmem']:3d} "
              f"salu={c['salu']:3d} wait={c['wait']:3d}  valu_cycles~{cyc}")
        print('             ' + ' '.join(f'{o}:{n}' for o, n in top))
    print('total', dict(tot))


main()
