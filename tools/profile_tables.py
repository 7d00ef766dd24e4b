"""Per-kernel table of one training step from four rocprofv3 runs (rocpd sqlite databases):

    python tools/profile_tables.py <kernel_trace.db> <steps_in_window> <window_ms> <fetch.db> <write.db> <mfma.db> <pmc_steps> <out_prefix>

  kernel_trace.db : rocprofv3 --kernel-trace -- python bench.py ...   (graph replays; durations come from here, last window_ms only)
  fetch/write.db  : rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --no-graph --steps S ...   (separate passes)
  mfma.db         : rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- same command
Counter conventions (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE in KB, FETCH doubled on gfx950; SQ_VALU_MFMA_BUSY_CYCLES counts
32 cycles per 32x32x16 (16 per 16x16x32) bf16 MFMA summed over all SIMDs, i.e. 1024 FLOP per counted cycle; MFMA utilisation of a
kernel = busy cycles / (1024 SIMDs x kernel cycles) = (busy x 1024 FLOP / duration) / 2.5 PFLOP/s.
Writes <out_prefix>_kernel_table.txt / .json."""
import json
import re
import sqlite3
import sys

PEAK_TF, HBM_TBS = 2500.0, 8.0


def short(n):
    n = re.sub(r'\(anonymous namespace\)::|phg::', '', n or ''); n = re.sub(r'\(.*\)$', '', n); n = re.sub(r'^void ', '', n)
    return n[:90]


def step_window(cur, steps, marker='ce_fwd_kernel'):
    """[t0, t1) covering exactly `steps` whole steps at the END of the trace: the marker kernel (one launch per step) delimits
    steps, so launches/step are integers for graph replays whatever the step time was on that box."""
    starts = [r[0] for r in cur.execute(f"select start from kernels where name like '%{marker}%' order by start")]
    if len(starts) < steps + 1:
        raise SystemExit(f'{marker}: only {len(starts)} launches in the trace, need {steps + 1}')
    return starts[-1 - steps], starts[-1]


def durations(db, steps, window_ms=None):
    cur = sqlite3.connect(db).cursor()
    t0, t1 = step_window(cur, int(steps))
    rows = cur.execute(f'select name, count(*), sum(end-start) from kernels where start >= {t0} and start < {t1} group by name')
    out = {}
    for n, c, t in rows:                                                          # launches/step, us/step (names that collide after shortening are summed)
        a = out.get(short(n), (0.0, 0.0)); out[short(n)] = (a[0] + c / steps, a[1] + t / steps / 1e3)
    out['__wall_ms_per_step__'] = (0, (t1 - t0) / steps / 1e6)
    return out


def counter(db, name, steps):
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
                where = ''
                if 'start' in cols:         # whole steps of the eager counter run only: between the first and the last once-per-step marker
                    marks = [r[0] for r in cur.execute(f'select distinct "start" from "{n}" where "{cn}" = ? and "{kn}" like \'%ce_fwd_kernel%\' order by "start"', (name,))]
                    if len(marks) >= 2:
                        where, steps = f' and "start" >= {marks[0]} and "start" < {marks[-1]}', float(len(marks) - 1)
                rows = list(cur.execute(f'select "{kn}", sum("{val}"), count(*) from "{n}" where "{cn}" = ?' + where + f' group by "{kn}"', (name,)))
            except sqlite3.Error:
                continue
            if rows:
                out = {}
                for k, v, c in rows:
                    a = out.get(short(k), (0.0, 0.0)); out[short(k)] = (a[0] + v / steps, a[1] + c / steps)
                return out
    raise SystemExit(f'{db}: counter {name} not found')


def main():
    kt, steps, window = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
    fdb, wdb, mdb, psteps, out = sys.argv[4], sys.argv[5], sys.argv[6], float(sys.argv[7]), sys.argv[8]
    dur = durations(kt, steps, window)
    wall_ms = dur.pop('__wall_ms_per_step__')[1]
    fetch, write = counter(fdb, 'FETCH_SIZE', psteps), counter(wdb, 'WRITE_SIZE', psteps)
    busy, gui = counter(mdb, 'SQ_VALU_MFMA_BUSY_CYCLES', psteps), counter(mdb, 'GRBM_GUI_ACTIVE', psteps)
    rows = []
    for k, (n, us) in dur.items():
        fb = 2 * fetch.get(k, (0, 0))[0] * 1024
        wb = write.get(k, (0, 0))[0] * 1024
        mf = busy.get(k, (0, 0))[0] * 1024.0                      # FLOP per step executed on the matrix cores
        rows.append(dict(kernel=k, launches_per_step=round(n, 1), us_per_step=round(us, 1), avg_us=round(us / max(n, 1e-9), 2),
                         hbm_gb_per_step=round((fb + wb) / 1e9, 3), hbm_tb_s=round((fb + wb) / (us * 1e-6) / 1e12, 3) if us else 0,
                         mfma_tflop_per_step=round(mf / 1e12, 4), mfma_tflops=round(mf / (us * 1e-6) / 1e12, 1) if us else 0,
                         mfma_util=round(mf / (us * 1e-6) / 1e12 / PEAK_TF, 4) if us else 0,
                         mfma_busy_over_gui_active=round(busy.get(k, (0, 0))[0] / max(gui.get(k, (1, 0))[0], 1), 3)))
    rows.sort(key=lambda r: -r['us_per_step'])
    tot_us = sum(r['us_per_step'] for r in rows)
    tot = dict(wall_ms_per_step=round(wall_ms, 3), kernel_us_per_step=round(tot_us, 1), launches_per_step=round(sum(r['launches_per_step'] for r in rows), 1),
               hbm_gb_per_step=round(sum(r['hbm_gb_per_step'] for r in rows), 2),
               mfma_tflop_per_step=round(sum(r['mfma_tflop_per_step'] for r in rows), 3))
    gemm = [r for r in rows if re.search(r'gemm_\w*kernel', r['kernel'])]
    gl = sum(r['launches_per_step'] for r in gemm)
    tot['gemm'] = dict(us_per_step=round(sum(r['us_per_step'] for r in gemm), 1), launches_per_step=round(gl, 1),
                       hbm_bytes_per_launch=round(sum(r['hbm_gb_per_step'] for r in gemm) * 1e9 / max(gl, 1)),
                       mfma_tflop_per_step=round(sum(r['mfma_tflop_per_step'] for r in gemm), 3))
    json.dump(dict(totals=tot, kernels=rows), open(out + '_kernel_table.json', 'w'), indent=1)
    with open(out + '_kernel_table.txt', 'w') as f:
        f.write(f'per step ({int(steps)} whole steps, {wall_ms:.2f} ms wall each): {tot_us / 1e3:.2f} ms of kernel time, {tot["launches_per_step"]:.0f} launches, HBM-side traffic {tot["hbm_gb_per_step"]:.1f} GB '
                f'(2 x FETCH_SIZE + WRITE_SIZE), {tot["mfma_tflop_per_step"]:.2f} TFLOP on the matrix cores (SQ_VALU_MFMA_BUSY_CYCLES x 1024)\n')
        f.write(f'{"kernel":70s} {"n/step":>7s} {"avg us":>8s} {"ms/step":>8s} {"TB/s":>6s} {"TF/s":>7s} {"MFMA util":>9s}\n')
        for r in rows[:40]:
            f.write(f'{r["kernel"][:70]:70s} {r["launches_per_step"]:7.1f} {r["avg_us"]:8.1f} {r["us_per_step"] / 1e3:8.3f} {r["hbm_tb_s"]:6.2f} '
                    f'{r["mfma_tflops"]:7.1f} {r["mfma_util"]:9.3f}\n')
    print(open(out + '_kernel_table.txt').read())


if __name__ == '__main__':
    main()
