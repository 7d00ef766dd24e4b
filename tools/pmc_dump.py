"""per-kernel averages of every counter in a rocprofv3 --pmc rocpd database:  python tools/pmc_dump.py <db> [name-regex]"""
import re
import sqlite3
import sys
from collections import defaultdict

db, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '.')
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
    if not (cn and val and kn and cn != kn):
        continue
    try:
        rows = list(cur.execute(f'select "{kn}", "{cn}", "{val}" from "{n}"'))
    except sqlite3.Error:
        continue
    if not rows:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for k, c, v in rows:
        if re.search(pat, k or ''):
            k = re.sub(r'\(anonymous namespace\)::|phg::|\(.*\)$|^void ', '', k)[:70]
            acc[(k, c)][0] += float(v); acc[(k, c)][1] += 1
    for (k, c), (t, m) in sorted(acc.items()):
        print(f'{k:70s} {c:28s} avg {t / m:14.1f}  n={m}')
    break
else:
    print('no counter view found in', db, names[:30])
