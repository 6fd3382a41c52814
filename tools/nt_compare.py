"""Side-by-side of two tools/nt_bench.py logs: per shape microseconds and the difference."""
import re, sys
def load(p):
    d = {}
    for line in open(p):
        m = re.match(r"\s*(\d+) pro=\s*(\d+) R=\s*(\d+) N=\s*(\d+) K=\s*(\d+):\s*([\d.]+) us", line)
        if m: d[int(m.group(1))] = (tuple(int(v) for v in m.groups()[1:5]), float(m.group(6)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
for k in sorted(a):
    if k not in b: continue
    (pro, R, N, K), ua = a[k]; ub = b[k][1]; ta += ua; tb += ub
    flag = " <<" if ub < 0.9 * ua else (" >>" if ub > 1.1 * ua else "")
    print(f"{k:3d} pro={pro:2d} R={R:7d} N={N:5d} K={K:5d}: {ua:8.1f} -> {ub:8.1f} us {ub - ua:+7.1f}{flag}")
print(f"TOTAL {ta:.1f} -> {tb:.1f}")
