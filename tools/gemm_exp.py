import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloud_bridge_amd import _lib
sig = _lib.SIGNATURES["pcb_gemm_nt_bf16"]
def timeit(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
variants = sys.argv[1:] or ["BASE", "E_NOEPI", "E_NOMFMA", "E_NOLOADA"]
for v in variants:
    L = ctypes.CDLL(os.path.join(ROOT, "exp_tmp", f"libexp_{v}.so"))
    L.pcb_gemm_nt_bf16.argtypes = sig
    for (pro, R, K, N, stats) in [(1, 524288, 64, 64, 1), (1, 524288, 64, 64, 0), (0, 524288, 64, 64, 1), (2, 524288, 128, 64, 0)]:
        x = torch.randn(R, K, device="cuda").to(torch.bfloat16); y = torch.randn(R, K, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, K, device="cuda").to(torch.bfloat16); out = torch.empty(R, N, device="cuda", dtype=torch.bfloat16)
        c = [torch.rand(K, device="cuda") for _ in range(4)]; sums = torch.zeros(2, N, device="cuda")
        f = lambda: L.pcb_gemm_nt_bf16(pro, x.data_ptr(), y.data_ptr(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(),
                                       0, 0, 16, 1, w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr() if stats else 0,
                                       torch.cuda.current_stream().cuda_stream)
        print(f"{v:10s} pro={pro} stats={stats} R={R} K={K} N={N}: {timeit(f):8.1f} us")
