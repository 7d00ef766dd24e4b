"""HBM traffic per GEMM launch from two rocprofv3 --pmc passes (rocpd sqlite databases):

    python tools/pmc_summary.py <fetch.db> <write.db> <steps> [out.json]

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B read requests
are tallied at 64 B for wide coalesced streams).  The rocpd schema differs between rocprofv3 builds, so the counter view is
found by introspection (a view/table that has a counter-name column, a value column and a kernel-name column).
"""
import json
import re
import sqlite3
import sys


def find_rows(db, counter, marker='ce_fwd_kernel'):
    """(view, rows, whole_steps): (kernel name, value) of every dispatch inside the whole steps of the run -- the dispatches between the
    first and the last launch of the once-per-step marker kernel, so that initialisation kernels and the partial first / last step are
    left out (when the view carries no timestamps: every dispatch, whole_steps = None)."""
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
        if cn and val and kn and cn != kn:
            try:
                where, steps = '', None
                if 'start' in cols:
                    marks = [r[0] for r in cur.execute(f'select distinct "start" from "{n}" where "{cn}" = ? and "{kn}" like ? order by "start"',
                                                       (counter, f'%{marker}%'))]
                    if len(marks) >= 2:
                        where, steps = f' and "start" >= {marks[0]} and "start" < {marks[-1]}', len(marks) - 1
                rows = list(cur.execute(f'select "{kn}", "{val}" from "{n}" where "{cn}" = ?' + where, (counter,)))
            except sqlite3.Error:
                continue
            if rows:
                return n, rows, steps
    raise SystemExit(f'{db}: no view with {counter} rows found; tables/views: {names}')


def per_kernel(rows, pat):
    tot, n, all_tot = 0.0, 0, 0.0
    for name, v in rows:
        all_tot += float(v)
        if re.search(pat, name or ''):
            tot += float(v); n += 1
    return tot, n, all_tot


def main():
    fdb, wdb, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
    pat = r'gemm_\w*kernel'
    fv, frows, fsteps = find_rows(fdb, 'FETCH_SIZE')
    wv, wrows, wsteps = find_rows(wdb, 'WRITE_SIZE')
    if fsteps and wsteps:
        steps = float(min(fsteps, wsteps))        # whole steps actually covered (the argument is the fallback)
    f_kb, f_n, f_all = per_kernel(frows, pat)
    w_kb, w_n, w_all = per_kernel(wrows, pat)
    n = max(f_n, 1)
    out = {
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes) -- python bench.py --steps 2 --warmup 1 '
                  '--no-graph --no-cpu-baseline --no-roofline',
        'note': 'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B for wide coalesced streams); '
                'WRITE_SIZE uncorrected; both counters are KB; dispatches of the whole steps of the run (between the first and the last ce_fwd_kernel)',
        'whole_steps': steps,
        'kernel': 'every gemm_*kernel<*> instance (128x128 / 64x64, grouped, intra-block k split, big-tile, big-tile grouped)',
        'views': [fv, wv],
        'launches_per_step': round(f_n / steps, 1),
        'fetch_bytes_per_launch_raw': f_kb * 1024 / n,
        'fetch_bytes_per_launch_corrected': 2 * f_kb * 1024 / n,
        'write_bytes_per_launch': w_kb * 1024 / max(w_n, 1),
        'hbm_bytes_per_launch': 2 * f_kb * 1024 / n + w_kb * 1024 / max(w_n, 1),
        'whole_step_hbm_gb': (2 * f_all + w_all) * 1024 / steps / 1e9,
    }
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
