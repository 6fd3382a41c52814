"""Micro-benchmark of pcb_gemm_tn_bf16 (weight gradient) at the shapes of one pn2_msg training step.

    PCB_TN_TRACE=1 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --exec eager 2> trace.txt
    python tools/tn_bench.py trace.txt            (PCB_TN_DEPTH / PCB_TN_TARGET select the kernel variant)

Every distinct launch (apro, bpro, R, M, N, colsum) of the LAST traced step runs back to back on rotating operand
sets; prints microseconds and TB/s of algorithmic bytes per shape (x its count in the step) and the step total.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import _lib  # noqa: E402


def shapes(path):
    rows = [tuple(int(v) for v in l.split()[1:]) for l in open(path) if l.startswith("[pcb_tn]")]
    return last_period(rows)


def last_period(rows):
    """The last step of a trace that holds several identical steps (the shortest period the tail repeats with)."""
    for p in range(8, len(rows) // 2 + 1):
        if rows[-p:] == rows[-2 * p:-p]:
            return rows[-p:]
    return rows


def main():
    L = _lib.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    ns = 16
    todo = shapes(sys.argv[1])
    count = {}
    for s in todo:
        count[s] = count.get(s, 0) + 1
    total = total_bytes = 0.0
    for (apro, bpro, R, M, N, colsum), cnt in count.items():
        per_set = R * M * 2 * (2 if apro >= 2 else 1) + R * N * 2
        nsets = max(2, min(8, int(400e6 // per_set) + 1))
        sets = [((torch.randn(R, M, device=dev) * 0.1).bfloat16(), (torch.randn(R, M, device=dev) * 0.1).bfloat16(),
                 (torch.randn(R, N, device=dev) * 0.1).bfloat16()) for _ in range(nsets)]
        v = [torch.rand(M, device=dev) + 0.5 for _ in range(4)]
        xv = [torch.rand(N, device=dev) + 0.5 for _ in range(2)]
        dout = torch.randn(R // ns, M, device=dev)
        arg = torch.randint(0, ns, (R // ns, M), device=dev, dtype=torch.uint8)
        ws = torch.empty(L.pcb_gemm_tn_workspace(R, M, N), dtype=torch.float32, device=dev)
        dW = torch.empty(M, N, dtype=torch.float32, device=dev)
        db = torch.empty(M, dtype=torch.float32, device=dev)

        def run(i):
            dz, y, x = sets[i % nsets]
            if colsum:
                rc = L.pcb_gemm_tn_bias_bf16(dz.data_ptr(), x.data_ptr(), R, M, N, ws.data_ptr(), dW.data_ptr(), N, 0, db.data_ptr(), st)
            else:
                rc = L.pcb_gemm_tn_bf16(apro, dz.data_ptr(), y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                                        v[3].data_ptr(), dout.data_ptr(), arg.data_ptr(), ns, 1, bpro, x.data_ptr(),
                                        xv[0].data_ptr(), xv[1].data_ptr(), 1, R, M, N, ws.data_ptr(), dW.data_ptr(), N, 0, st)
            assert rc == 0, rc

        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 6 * nsets
        e0.record()
        for i in range(n):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        byt = 2 * R * N + (2 * R * M if apro < 2 else 4 * R * M if apro == 2 else 2 * R * M + 5 * (R // ns) * M)
        total += us * cnt
        total_bytes += byt * cnt
        print(f"apro={apro} bpro={bpro} R={R:7d} M={M:5d} N={N:5d} bias={colsum} x{cnt}: {us:8.1f} us {byt / us / 1e6:6.2f} TB/s (incl. slab sums)")
        del sets
    print(f"TOTAL {total:9.1f} us  {total_bytes / total / 1e6:6.2f} TB/s  depth={os.environ.get('PCB_TN_DEPTH', '1')} target={os.environ.get('PCB_TN_TARGET', '512')}")


if __name__ == "__main__":
    main()
