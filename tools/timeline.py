"""Ordered kernel timeline of ONE step from a `rocprofv3 --kernel-trace --output-format csv` run:
    python tools/timeline.py <dir with *_kernel_trace.csv> [step index from the end, default 2] > timeline.txt
One line per launch: start (us from the step's first kernel), queue id, duration, idle gap on ITS queue
before it, grid/block, short kernel name.  A step boundary = the fused-Adam kernel (multi_tensor_apply)."""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:100]


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
    lo, hi = ends[-back - 1] + 1, ends[-back] + 1
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    last_end = {}
    busy = {}
    print(f"# {len(step)} launches, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us from first start to last end")
    for r in step:
        q = r["Queue_Id"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        busy[q] = busy.get(q, 0) + e - s
        grid = f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}/{r['Workgroup_Size_X']}"
        print(f"{(s - t0) / 1e3:9.1f} q{q:>2} {(e - s) / 1e3:8.1f} gap {gap:7.1f} {grid:>16} {short(r['Kernel_Name'])}")
    for q, b in busy.items():
        print(f"# queue {q}: busy {b / 1e3:.1f} us")


main()
