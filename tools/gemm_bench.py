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
    v = [torch.rand(K, device="cuda") for _ in range(4)]; sums = torch.zeros(L.pcb_gemm_nt_partials(pro, R, N), 2, N, device="cuda")
    dout = torch.randn(R // ns, K, device="cuda"); arg = torch.randint(0, ns, (R // ns, K), device="cuda", dtype=torch.uint8)
    f = lambda: L.pcb_gemm_nt_bf16(pro, x.data_ptr(), y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                   dout.data_ptr(), arg.data_ptr(), ns, 1, w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr() if stats else 0, st())
    us = timeit(f)
    byt = 2 * R * N + (2 * R * K if pro < 2 else 4 * R * K if pro == 2 else 2 * R * K + 5 * (R // ns) * K)
    print(f"nt pro={pro} stats={stats} R={R} K={K} N={N}: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s")
def ntred(pro, R, K, N, ns=16):
    x = torch.randn(R, K, device="cuda").to(torch.bfloat16); y = torch.randn(R, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16); out = torch.empty(R, N, device="cuda", dtype=torch.bfloat16)
    yp = torch.randn(R, N, device="cuda").to(torch.bfloat16); rv = [torch.rand(N, device="cuda") for _ in range(4)]
    v = [torch.rand(K, device="cuda") for _ in range(4)]; sums = torch.zeros(L.pcb_gemm_nt_partials(pro, R, N), 2, N, device="cuda")
    dout = torch.randn(R // ns, K, device="cuda"); arg = torch.randint(0, ns, (R // ns, K), device="cuda", dtype=torch.uint8)
    f = lambda: L.pcb_gemm_nt_red_bf16(pro, x.data_ptr(), y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                   dout.data_ptr(), arg.data_ptr(), ns, 1, w.data_ptr(), R, N, K, out.data_ptr(), yp.data_ptr(),
                                   rv[0].data_ptr(), rv[1].data_ptr(), rv[2].data_ptr(), rv[3].data_ptr(), 1, sums.data_ptr(), st())
    us = timeit(f)
    byt = 4 * R * N + (4 * R * K if pro == 2 else 2 * R * K + 5 * (R // ns) * K)
    print(f"nt RED pro={pro} R={R} K={K} N={N}: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s")
for args in [(2, 524288, 64, 64), (2, 524288, 128, 64), (3, 524288, 128, 64), (3, 262144, 256, 128), (2, 262144, 128, 128)]:
    ntred(*args)
for args in [(0, 524288, 8, 64, 1), (1, 524288, 64, 64, 1), (1, 524288, 64, 128, 1), (2, 524288, 64, 64, 0), (2, 524288, 128, 64, 0),
             (3, 524288, 128, 64, 0), (2, 262144, 128, 264, 0), (2, 262144, 256, 264, 0), (1, 262144, 264, 128, 1), (0, 262144, 264, 128, 1)]:
    nt(*args)

def tn(apro, bpro, R, M, N, ns=16):
    dz = torch.randn(R, M, device="cuda").to(torch.bfloat16); y = torch.randn(R, M, device="cuda").to(torch.bfloat16)
    xx = torch.randn(R, N, device="cuda").to(torch.bfloat16)
    v = [torch.rand(M, device="cuda") for _ in range(4)]; xv = [torch.rand(N, device="cuda") for _ in range(2)]
    dout = torch.randn(R // ns, M, device="cuda"); arg = torch.randint(0, ns, (R // ns, M), device="cuda", dtype=torch.uint8)
    ws = torch.empty(L.pcb_gemm_tn_workspace(R, M, N), device="cuda"); dw = torch.empty(M, N, device="cuda")
    f = lambda: L.pcb_gemm_tn_bf16(apro, dz.data_ptr(), y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                   dout.data_ptr(), arg.data_ptr(), ns, 1, bpro, xx.data_ptr(), xv[0].data_ptr(), xv[1].data_ptr(), 1, R, M, N,
                                   ws.data_ptr(), dw.data_ptr(), 0, 0, st())
    us = timeit(f)
    byt = 2 * R * N + (2 * R * M if apro == 0 else 4 * R * M if apro == 2 else 2 * R * M + 5 * (R // ns) * M)
    print(f"tn apro={apro} bpro={bpro} R={R} M={M} N={N}: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s  ws={ws.numel()*4/1e6:.1f}MB")
for args in [(2, 0, 524288, 64, 8), (2, 1, 524288, 64, 64), (3, 1, 524288, 128, 64), (2, 0, 262144, 128, 264), (2, 1, 262144, 128, 128),
             (3, 1, 262144, 256, 128), (2, 1, 262144, 128, 256), (2, 0, 262144, 256, 264), (0, 0, 262144, 264, 64)]:
    tn(*args)
