"""Aggregate a tools/timeline.py listing by kernel name: python tools/tl_agg.py <timeline.txt> [top]"""
import collections, re, sys
rows = [l for l in open(sys.argv[1]) if not l.startswith('#')]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for l in rows:
    m = re.match(r'\s*([\d.]+) q\s*(\d+)\s+([\d.]+) gap\s+([-\d.]+)\s+(\S+) (.*)', l)
    if not m:
        continue
    start, q, dur, gap, grid, name = m.groups()
    name = re.sub(r'[<(].*', '', name)[:44]
    a = agg[name]; a[0] += 1; a[1] += float(dur); tot += float(dur)
print(len(rows), 'launches, sum of durations', round(tot, 1), 'us')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1]:8.1f} us {v[0]:4d}  {k}")
