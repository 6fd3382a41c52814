"""Outputs and statistics slabs of pcb_gemm_nt_bf16 (plain prologue) against an fp32 torch product of the same
bf16 operands -- a quick check for kernel experiments selected by environment variables."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import _lib
L = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for R, N, K, stats in [(4096, 128, 64, 1), (5000, 128, 128, 1), (262144, 64, 64, 1), (70000, 256, 256, 0), (1000, 264, 64, 1), (333, 16, 128, 0)]:
    x = torch.randn(R, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16)
    out = torch.full((R, N), 7.0, device=dev, dtype=torch.bfloat16)
    nparts = int(os.environ.get("NT_NPARTS", "0")) or L.pcb_gemm_nt_partials(0, R, N)
    sums = torch.full((nparts, 2, N), 3.0, device=dev)
    rc = L.pcb_gemm_nt_bf16(0, x.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 1, 1, w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr() if stats else 0, nparts, st)
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t()
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    msg = f"R={R} N={N} K={K}: rc {rc} out err {err:.2e}"
    if stats:
        ob = out.float()
        s1 = sums[:, 0].sum(0); s2 = sums[:, 1].sum(0)
        msg += f" sum err {float((s1 - ob.sum(0)).abs().max() / ob.sum(0).abs().max().clamp_min(1)):.2e} sumsq err {float((s2 - (ob * ob).sum(0)).abs().max() / (ob * ob).sum(0).abs().max()):.2e}"
    print(msg)
