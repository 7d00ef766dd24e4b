"""FETCH_SIZE / WRITE_SIZE per launch of isolated, byte-countable kernels (calibration of the gfx950 counters) and of the GEMM shape
classes of the step -- the accounting behind `roofline.traffic` (round-3 review: 2-2.8x the algorithmic bytes, unexplained).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d OUT/f -o pmc -- python tools/fetch_probe.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d OUT/w -o pmc -- python tools/fetch_probe.py run
    python tools/fetch_probe.py table OUT/f/.../pmc_results.db OUT/w/.../pmc_results.db

`run` launches every case REPS times back to back after a cache flush (768 MB fill), nothing else of this library in between; `table`
walks the counter rows in dispatch order and assigns them to the same case list."""
import os
import re
import sqlite3
import sys

REPS = 3
BF_BYTES = 2


def cases():
    """(label, kind, M, N, K, kwargs, algorithmic bytes read, written)"""
    out = []
    n = 64 << 20
    out.append(('add_bf16 64M elements (calibration: 2 x 128 MB read, 128 MB written)', 'add', n, 0, 0, {}, 2 * n * 2, n * 2))
    out.append(('cast f32->bf16 64M (256 MB read, 128 MB written)', 'cast', n, 0, 0, {}, n * 4, n * 2))

    def g(label, M, N, K, mode, **kw):
        rd = (M * K + N * K) * 2 + (M * N * 2 if kw.get('residual') else 0) + (M * N * 2 if kw.get('act_in') else 0)
        wr = M * N * 2 * (2 if kw.get('pre') else 1)
        out.append((label, 'gemm', M, N, K, dict(kw, mode=mode), rd, wr))
    for mode, tag in ((6, '256x128 DMA'), (0, '128x128 reg')):
        g(f'{tag}: out-proj 8320x768x768 +bias +res', 8320, 768, 768, mode, residual=True)
        g(f'{tag}: plain 8320x768x768', 8320, 768, 768, mode, bias=False)
        g(f'{tag}: dgrad 8320x768x768 [K,N]', 8320, 768, 768, mode, bias=False, tb=True)
        g(f'{tag}: proj 8320x768x3072 +res', 8320, 768, 3072, mode, residual=True)
        g(f'{tag}: c_fc 8192x3072x768 qgelu + derivative', 8192, 3072, 768, mode, act=1, pre=True)
        g(f'{tag}: qkv 8320x2304x768', 8320, 2304, 768, mode)
        g(f'{tag}: resampler kv 39680x1536x768', 39680, 1536, 768, mode)
    return out


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from prismer_amd import _lib, ops
    BF = torch.bfloat16
    flush = torch.empty(768 << 20, dtype=torch.uint8, device='cuda')
    torch.manual_seed(0)
    for label, kind, M, N, K, kw, rd, wr in cases():
        if kind == 'add':
            a = torch.randn(M, device='cuda').to(BF); b = torch.randn(M, device='cuda').to(BF); o = torch.empty_like(a)
            call = lambda: ops.add(a, b, out=o)
        elif kind == 'cast':
            a = torch.randn(M, device='cuda'); o = torch.empty(M, dtype=BF, device='cuda')
            call = lambda: ops.cast_to_bf16(a, out=o)
        else:
            _lib.lib.ph_gemm_tuning(kw['mode'], 128)
            a = (torch.randn(M, K, device='cuda') * 0.5).to(BF)
            w = (torch.randn(K, N, device='cuda') * 0.05).to(BF) if kw.get('tb') else (torch.randn(N, K, device='cuda') * 0.05).to(BF)
            args = {}
            if kw.get('tb'):
                args['trans_b'] = True
            if kw.get('bias', True):
                args['bias'] = torch.zeros(N, device='cuda')
            if kw.get('residual'):
                args['residual'] = torch.randn(M, N, device='cuda').to(BF)
            if kw.get('act'):
                args['act'] = kw['act']
            if kw.get('pre'):
                args['pre_out'] = torch.empty(M, N, dtype=BF, device='cuda'); args['pre_grad'] = True
            o = torch.empty(M, N, dtype=BF, device='cuda')
            call = lambda: ops.gemm(a, w, out=o, **args)
        torch.cuda.synchronize()
        for _ in range(REPS):
            flush.fill_(1)
            call()
        torch.cuda.synchronize()
    _lib.lib.ph_gemm_tuning(*_lib.GEMM_BIG_DEFAULT)


def rows_of(db, counter):
    cur = sqlite3.connect(db).cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    for n in names:
        try:
            cols = [r[1] for r in cur.execute(f'pragma table_info("{n}")')]
        except sqlite3.Error:
            continue
        cn = next((c for c in cols if c.lower() in ('counter_name', 'pmc_name', 'name_counter')), None)
        val = next((c for c in cols if c.lower() in ('value', 'counter_value')), None)
        kn = next((c for c in cols if c.lower() in ('kernel_name', 'name', 'kernel')), None)
        order = next((c for c in cols if c.lower() in ('dispatch_id', 'start', 'id')), None)
        if cn and val and kn and cn != kn and order:
            try:
                return list(cur.execute(f'select "{kn}", "{val}" from "{n}" where "{cn}" = ? order by "{order}"', (counter,)))
            except sqlite3.Error:
                continue
    raise SystemExit(f'no counter view in {db}: {names[:20]}')


def table(fdb, wdb):
    keep = re.compile(r'gemm|add_kernel|cast_f2b')
    f = [(k, v) for k, v in rows_of(fdb, 'FETCH_SIZE') if keep.search(k or '') and 'vectorized' not in k and 'elementwise' not in k]
    w = [(k, v) for k, v in rows_of(wdb, 'WRITE_SIZE') if keep.search(k or '') and 'vectorized' not in k and 'elementwise' not in k]
    cs = cases()
    print(f'{len(f)} FETCH rows, {len(w)} WRITE rows, {len(cs)} cases x {REPS} launches (the tail split makes two launches of a call with M = 8320 rows and a wide N)')
    print('FETCH_SIZE / WRITE_SIZE are KB as reported (no correction applied); algorithmic MB in brackets')
    i = j = 0
    for label, kind, M, N, K, kw, rd, wr in cs:
        # a call may be split into several launches (tail split): consume launches until REPS calls are covered -- here simply by name change
        per = 2 if (kind == 'gemm' and M == 8320 and N >= 2304 and kw['mode'] != 0 and False) else 1
        nf = REPS * per
        fs = f[i:i + nf]; ws = w[j:j + nf]; i += nf; j += nf
        fk = sum(float(v) for _, v in fs) / REPS; wk = sum(float(v) for _, v in ws) / REPS
        names = sorted({re.sub(r'\(.*', '', k)[-60:] for k, _ in fs})
        print(f'{label:62s} FETCH {fk / 1024:8.1f} MB [{rd / 1e6:7.1f}]  x2 = {2 * fk / 1024 / (rd / 1e6 / 1.048576):5.2f} of alg.   WRITE {wk / 1024:8.1f} MB [{wr / 1e6:7.1f}] = {wk / 1024 / (wr / 1e6 / 1.048576):5.2f}   {names}')


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    else:
        table(sys.argv[2], sys.argv[3])
