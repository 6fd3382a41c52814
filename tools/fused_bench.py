"""Micro-benchmark of pcb_bwd_fused_bf16 (one-pass layer backward: dx + dW + the sums of the layer below) at the shapes of
one pn2_msg training step, beside the two-kernel form it replaces (pcb_gemm_nt_red_bf16 + pcb_gemm_tn_bf16).

    PCB_NT_TRACE=1 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --exec eager 2> trace.txt
    python tools/fused_bench.py trace.txt        (lines `[pcb_nt] 22|23 R K C`: tag = prologue + 20)

Algorithmic bytes of a launch: dz + y read once (2 R C each; pooled form: y only + dout), x read once (2 R K), dx written
(2 R K).  Operand sets rotate over more than the 256 MB Infinity Cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import _lib  # noqa: E402


def shapes(path):
    rows = []
    for line in open(path):
        if line.startswith("[pcb_nt]"):
            tag, R, K, C = (int(v) for v in line.split()[1:5])
            if tag in (22, 23):
                rows.append((tag - 20, R, C, K))
    for p in range(2, len(rows) // 2 + 1):
        if rows[-p:] == rows[-2 * p:-p]:
            return rows[-p:]
    return rows


def main():
    L = _lib.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    todo = shapes(sys.argv[1]) if len(sys.argv) > 1 else [(2, 262144, 128, 128), (3, 262144, 256, 128), (2, 524288, 64, 64), (3, 524288, 128, 64)]
    count = {}
    for s in todo:
        count[s] = count.get(s, 0) + 1
    tot_f = tot_2 = 0.0
    ns = 32
    for (pro, R, C, K), cnt in count.items():
        per_set = R * C * 4 + R * K * 4
        nsets = max(2, min(8, int(600e6 // per_set) + 1))
        sets = [((torch.randn(R, C, device=dev) * 0.1).bfloat16(), (torch.randn(R, C, device=dev) * 0.1).bfloat16(),
                 (torch.randn(R, K, device=dev) * 0.1).bfloat16(), torch.empty(R, K, dtype=torch.bfloat16, device=dev))
                for _ in range(nsets)]
        v = [torch.rand(C, device=dev) + 0.5 for _ in range(4)]
        xv = [torch.rand(K, device=dev) + 0.5 for _ in range(4)]
        wt = (torch.randn(K, C, device=dev) * 0.1).bfloat16()
        dout = torch.randn(R // ns, C, device=dev)
        arg = torch.randint(0, ns, (R // ns, C), device=dev, dtype=torch.uint8)
        grid = min(512, (R + 63) // 64)
        red = torch.empty(grid, 2, K, device=dev)
        ws = torch.empty(grid * C * K, device=dev)
        dW = torch.empty(C, K, device=dev)
        nparts = L.pcb_gemm_nt_partials(pro, R, K)
        red2 = torch.empty(nparts, 2, K, device=dev)
        ws2 = torch.empty(L.pcb_gemm_tn_workspace(R, C, K), device=dev)

        def a_args(i):
            dz, y, x, dx = sets[i % nsets]
            return (dz.data_ptr() if pro == 2 else 0, y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                    dout.data_ptr() if pro == 3 else 0, arg.data_ptr() if pro == 3 else 0, ns, 1), x, dx

        def fused(i):
            a, x, dx = a_args(i)
            rc = L.pcb_bwd_fused_bf16(pro, *a, wt.data_ptr(), x.data_ptr(), xv[0].data_ptr(), xv[1].data_ptr(), xv[2].data_ptr(),
                                      xv[3].data_ptr(), 1, R, C, K, dx.data_ptr(), red.data_ptr(), grid, ws.data_ptr(), dW.data_ptr(), K, 0, st)
            assert rc == 0, rc

        def two(i):
            a, x, dx = a_args(i)
            rc = L.pcb_gemm_nt_red_bf16(pro, *a, wt.data_ptr(), R, K, C, dx.data_ptr(), x.data_ptr(), xv[0].data_ptr(), xv[1].data_ptr(),
                                        xv[2].data_ptr(), xv[3].data_ptr(), 1, red2.data_ptr(), nparts, st)
            rc |= L.pcb_gemm_tn_bf16(pro, *a, 1, x.data_ptr(), xv[0].data_ptr(), xv[1].data_ptr(), 1, R, C, K, ws2.data_ptr(), dW.data_ptr(), K, 0, st)
            assert rc == 0, rc

        res = []
        for fn in (fused, two):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 12
            a.record()
            for i in range(reps):
                fn(i)
            b.record()
            torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / reps * 1e3)
        byt = (2.0 * R * C * (2 if pro == 2 else 1) + (4.0 * (R // ns) * C if pro == 3 else 0) + 4.0 * R * K)
        tot_f += res[0] * cnt
        tot_2 += res[1] * cnt
        print(f"pro={pro} R={R:7d} C={C:4d} K={K:4d} x{cnt}: fused {res[0]:7.1f} us {byt / res[0] / 1e6:5.2f} TB/s | "
              f"gemm_nt_red + gemm_tn {res[1]:7.1f} us (incl. slab sums)")
    print(f"TOTAL fused {tot_f:.1f} us, two-kernel form {tot_2:.1f} us")


if __name__ == "__main__":
    main()
