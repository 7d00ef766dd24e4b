"""Summarise a PH_PROF_DUMP csv (family, ms, flops, desc per launch; bench.py's instrumented eager pass): per description
total ms/step, launches/step, average us, TF/s.   python tools/percall.py dump.csv steps [top]"""
import collections
import sys

path, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
tot = 0.0
for line in open(path):
    fam, ms, fl, desc = line.rstrip('\n').split(',', 3)
    key = f'fam{fam} {desc}'
    a = acc[key]
    a[0] += float(ms); a[1] += 1; a[2] += float(fl)
    tot += float(ms)
print(f'total {tot / steps:.2f} ms/step')
for key, (ms, n, fl) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f'{ms / steps:8.3f} ms/step  n={n / steps:5.0f}  avg {ms / n * 1e3:8.1f} us  {fl / ms / 1e9 if ms > 0 else 0:7.1f} TF  {key}')
