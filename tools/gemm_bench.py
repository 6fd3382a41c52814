"""Scratch micro-benchmark of the fused GEMM entry points at the shapes of the bench step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import _lib
L = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
def nt(pro, R, K, N, stats, ns=16):
    x = torch.randn(R, K, device="cuda").to(torch.bfloat16); y = torch.randn(R, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16); out = torch.empty(R, N, device="cuda", dtype=torch.bfloat16)
    v = [torch.rand(K, device="cuda") for _ in range(4)]; sums = torch.zeros(2, N, device="cuda")
    dout = torch.randn(R // ns, K, device="cuda"); arg = torch.randint(0, ns, (R // ns, K), device="cuda", dtype=torch.uint8)
    f = lambda: L.pcb_gemm_nt_bf16(pro, x.data_ptr(), y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                   dout.data_ptr(), arg.data_ptr(), ns, 1, w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr() if stats else 0, st())
    us = timeit(f)
    byt = 2 * R * N + (2 * R * K if pro < 2 else 4 * R * K if pro == 2 else 2 * R * K + 5 * (R // ns) * K)
    print(f"nt pro={pro} stats={stats} R={R} K={K} N={N}: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s")
for args in [(0, 524288, 8, 64, 1), (1, 524288, 64, 64, 1), (1, 524288, 64, 128, 1), (2, 524288, 64, 64, 0), (2, 524288, 128, 64, 0),
             (3, 524288, 128, 64, 0), (2, 262144, 128, 264, 0), (2, 262144, 256, 264, 0), (1, 262144, 264, 128, 1), (0, 262144, 264, 128, 1)]:
    nt(*args)
