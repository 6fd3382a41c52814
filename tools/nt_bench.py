"""Micro-benchmark of the bf16 gemm_nt entry points at the shapes of one pn2_msg training step.

The list below is the launch trace of one step (PCB_TIMER_VERBOSE=1 python bench.py --steps 1): tag = pro,
+10 for the RED epilogue.  Each shape runs on rotating operand sets (larger than the 256 MB Infinity Cache
together) so that the rate is the streaming one of the real step.  Prints microseconds, TB/s of algorithmic
bytes per shape and the total: the figure a kernel variant has to lower.

    python tools/nt_bench.py [--only N] [--fp32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import _lib  # noqa: E402

STEP = """0 262144 64 8|1 262144 64 64|1 262144 128 64|0 524288 64 8|1 524288 64 64|1 524288 128 64|1 131072 128 128|
1 131072 256 128|1 262144 128 128|1 262144 256 128|1 32768 256 256|1 32768 512 256|1 65536 256 256|1 65536 512 256|
0 8192 384 1536|0 8192 1536 384|0 8192 16 8|0 8192 256 16|0 8192 1024 1536|1 8192 256 1024|0 16384 128 512|0 16384 512 128|
0 16384 16 8|0 16384 256 16|0 16384 256 512|1 16384 256 256|0 262144 64 264|0 262144 264 64|0 262144 16 8|0 262144 128 16|
0 262144 256 264|1 262144 128 256|0 8192 128 256|0 16384 128 256|0 262144 128 128|0 262144 128 384|0 262144 8 128|
0 262144 128 8|2 262144 384 128|2 262144 128 128|2 16384 256 128|2 8192 256 128|2 262144 256 128|2 262144 264 256|
0 262144 16 128|0 262144 64 264|2 262144 264 64|2 16384 256 256|2 16384 512 256|0 16384 16 256|0 16384 128 512|
2 16384 512 128|2 8192 1024 256|2 8192 1536 1024|0 8192 16 256|0 8192 384 1536|2 8192 1536 384|3 65536 256 512|
2 65536 256 256|0 8192 512 256|3 32768 256 512|2 32768 256 256|0 8192 512 256|13 262144 128 256|12 262144 128 128|
0 16384 256 128|13 131072 128 256|12 131072 128 128|0 16384 256 128|13 524288 64 128|12 524288 64 64|13 262144 64 128|
12 262144 64 64"""


def shapes(trace=None):
    """The built-in list, or the launches of a trace: `PCB_NT_TRACE=1 python bench.py --steps 1 --warmup 0 --no-extras
    --no-cpu-baseline --exec eager 2> trace.txt` (lines `[pcb_nt] tag R N K`; tag 20 = fp32 output, run as a plain GEMM).
    The first quarter of a trace is the probe / warm-up free single step as the bench enqueues it."""
    out = []
    if trace:
        for line in open(trace):
            if line.startswith("[pcb_nt]"):
                tag, R, N, K = (int(v) for v in line.split()[1:5])
                if R > 0 and tag <= 20:
                    out.append((0 if tag == 20 else tag, R, N, K))
        for p in range(8, len(out) // 2 + 1):   # several identical steps: the last one
            if out[-p:] == out[-2 * p:-p]:
                return out[-p:]
        return out
    for item in STEP.replace("\n", "").split("|"):
        pro, R, N, K = (int(v) for v in item.split())
        out.append((pro, R, N, K))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace", nargs="?", default=None, help="launch trace (see shapes()); default: the built-in list")
    ap.add_argument("--only", default="", help="comma-separated indices into the list")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--min-us", type=float, default=0.0, help="list only shapes slower than this")
    a = ap.parse_args()
    L = _lib.load()
    only = {int(v) for v in a.only.split(",") if v}
    dev = torch.device("cuda", 0)
    dt = torch.float32 if a.fp32 else torch.bfloat16
    es = 4 if a.fp32 else 2
    sfx = "f32" if a.fp32 else "bf16"
    nt = getattr(L, "pcb_gemm_nt_" + sfx)
    ntred = getattr(L, "pcb_gemm_nt_red_" + sfx)
    st = torch.cuda.current_stream().cuda_stream
    ns = 16
    total = total_bytes = 0.0
    rows = []
    for idx, (tag, R, N, K) in enumerate(shapes(a.trace)):
        if only and idx not in only:
            continue
        red, pro = tag >= 10, tag % 10
        per_set = R * K * es * (2 if pro >= 2 else 1) + R * N * es * (2 if red else 1)
        nsets = max(2, min(8, int(400e6 // per_set) + 1))
        sets = []
        for _ in range(nsets):
            x = torch.randn(R, K, device=dev).to(dt)
            y = torch.randn(R, K, device=dev).to(dt) if pro >= 2 else x
            out = torch.empty(R, N, device=dev, dtype=dt)
            yp = torch.randn(R, N, device=dev).to(dt) if red else out
            sets.append((x, y, out, yp))
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        v = [torch.rand(K, device=dev) + 0.5 for _ in range(4)]
        rv = [torch.rand(N, device=dev) + 0.5 for _ in range(4)]
        stats = (pro <= 1 and N >= 32) or red
        nparts = int(os.environ.get("NT_NPARTS", "0")) or L.pcb_gemm_nt_partials(pro, R, N)
        sums = torch.zeros(nparts, 2, N, device=dev)
        dout = torch.randn(R // ns, K, device=dev)
        arg = torch.randint(0, ns, (R // ns, K), device=dev, dtype=torch.uint8)

        def run(i):
            x, y, out, yp = sets[i % nsets]
            common = (pro, x.data_ptr(), y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                      dout.data_ptr(), arg.data_ptr(), ns, 1, w.data_ptr(), R, N, K, out.data_ptr())
            if red:
                rc = ntred(*common, yp.data_ptr(), rv[0].data_ptr(), rv[1].data_ptr(), rv[2].data_ptr(), rv[3].data_ptr(), 1,
                           sums.data_ptr(), nparts, st)
            else:
                rc = nt(*common, sums.data_ptr() if stats else 0, nparts, st)
            assert rc == 0, rc

        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = a.reps * nsets
        e0.record()
        for i in range(n):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        byt = es * R * N * (2 if red else 1) + (es * R * K if pro < 2 else 2 * es * R * K if pro == 2 else es * R * K + 5 * (R // ns) * K)
        total += us
        total_bytes += byt
        rows.append((idx, tag, R, N, K, us, byt / us / 1e6))
        del sets
    for idx, tag, R, N, K, us, tbs in rows:
        if us >= a.min_us:
            print(f"{idx:3d} pro={tag:2d} R={R:7d} N={N:5d} K={K:5d}: {us:8.1f} us {tbs:6.2f} TB/s")
    print(f"TOTAL {total:9.1f} us  {total_bytes / 1e6:9.1f} MB  {total_bytes / total / 1e6:6.2f} TB/s  (back-to-back launches: "
          "includes the launch gaps the step hides less well)")


if __name__ == "__main__":
    main()
