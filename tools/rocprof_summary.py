"""kernel-trace summary from a rocprofv3 rocpd sqlite database:
    python tools/rocprof_summary.py <db> [out.csv|-] [steps] [window_ms]
window_ms restricts the summary to kernels that started in the last window_ms of the trace (steady-state graph replays)."""
import csv, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
where = ""
if len(sys.argv) > 4:
    tmax = list(cur.execute("select max(end) from kernels"))[0][0]
    where = f" where start >= {tmax - int(float(sys.argv[4]) * 1e6)}"
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels" + where + " group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*\)$', '', n); n = re.sub(r'^void ', '', n)
    return n[:100]
if len(sys.argv) > 2 and sys.argv[2] != '-':
    with open(sys.argv[2], 'w', newline='') as f:
        w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage'])
        for r in rows: w.writerow([short(r[0]), r[1], int(r[2]), round(r[3], 1), r[4], r[5], round(100 * r[2] / tot, 3)])
print(f'total {tot/1e6:.1f} ms over {steps:g} steps = {tot/1e6/steps:.2f} ms/step, {sum(r[1] for r in rows)} dispatches')
for r in rows[:40]:
    print(f'{100*r[2]/tot:6.2f}%  {r[2]/1e6/steps:7.3f} ms/step  n/step={r[1]/steps:7.1f} avg {r[3]/1e3:8.1f} us  {short(r[0])}')
