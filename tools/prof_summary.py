"""Condense a rocprofv3 kernel_stats.csv: top kernels by time, per-step figures."""
import csv, sys, glob
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms, per step {tot/1e6/steps:.3f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    n = r["Name"]
    n = n[:90]
    print(f"{float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step {int(r['Calls'])/steps:7.1f} calls/step {float(r['AverageNs'])/1e3:9.1f} us avg  {n}")
