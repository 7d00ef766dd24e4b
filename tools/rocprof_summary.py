"""kernel-trace summary from a rocprofv3 rocpd sqlite database:
    python tools/rocprof_summary.py <db> [out.csv|-] [steps] [window_ms]
With a 4th argument the summary covers exactly the last <steps> whole steps of the trace (steady-state graph replays)."""
import csv, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
where = ""
if len(sys.argv) > 4:      # whole steps at the end of the trace, delimited by the once-per-step ce_fwd_kernel (window_ms is ignored)
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from profile_tables import step_window
    t0, t1 = step_window(cur, int(float(sys.argv[3])))
    where = f" where start >= {t0} and start < {t1}"
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels" + where + " group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
def short(n):
    n = re.sub(r'\(anonymous namespace\)::|phg::', '', n); n = re.sub(r'\(.*\)$', '', n); n = re.sub(r'^void ', '', n)
    return n[:100]
if len(sys.argv) > 2 and sys.argv[2] != '-':
    with open(sys.argv[2], 'w', newline='') as f:
        w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage'])
        for r in rows: w.writerow([short(r[0]), r[1], int(r[2]), round(r[3], 1), r[4], r[5], round(100 * r[2] / tot, 3)])
print(f'total {tot/1e6:.1f} ms over {steps:g} steps = {tot/1e6/steps:.2f} ms/step, {sum(r[1] for r in rows)} dispatches')
for r in rows[:40]:
    print(f'{100*r[2]/tot:6.2f}%  {r[2]/1e6/steps:7.3f} ms/step  n/step={r[1]/steps:7.1f} avg {r[3]/1e3:8.1f} us  {short(r[0])}')

# GEMM launches by kernel and grid (one grid size ~ one problem shape class): in-situ per-shape durations of the same window
try:
    rows = list(cur.execute("select name, grid_x / workgroup_x, count(*), avg(end-start), min(end-start), max(end-start) from kernels"
                            + (where + " and" if where else " where") + " name like '%gemm%' group by name, grid_x order by sum(end-start) desc limit 40"))
    print('\nGEMM launches by kernel and grid (blocks):')
    for n, g, c, a, mn, mx in rows:
        print(f'  n/step={c/steps:6.1f}  blocks={int(g):6d}  avg {a/1e3:7.1f} us  min {mn/1e3:7.1f}  max {mx/1e3:7.1f}  {short(n)}')
except sqlite3.Error as e:
    print('by-grid table unavailable:', e)
