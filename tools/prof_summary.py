"""summarise a PH_PROF_DUMP csv: top call descriptions by total time"""
import sys, collections
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for line in open(sys.argv[1]):
    fam, ms, fl, desc = line.rstrip('\n').split(',', 3)
    a = agg[(fam, desc)]; a[0] += float(ms); a[1] += 1; a[2] += float(fl)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print(f'total {tot/steps:.2f} ms/step')
for (fam, desc), (ms, n, fl) in rows[:45]:
    tf = fl / ms / 1e9 if ms > 0 and fl > 0 else 0
    print(f'{ms/steps:8.3f} ms/step  n={n//steps:4d}  avg {ms/n*1e3:8.1f} us  {tf:7.1f} TF  fam{fam} {desc}')
