"""Scratch: library GEMM (torch.matmul -> hipBLASLt) time at the compute-shaped layer shapes of the PN2-MSG step, as a yardstick
for the own kernels (profiles/r03_gemm_nt_shapes_final.txt / r03_gemm_tn_shapes_final.txt)."""
import torch
dev = "cuda"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (R, N, K) in [(8192, 1536, 1024), (8192, 1024, 1536), (8192, 384, 1536), (8192, 1536, 384), (8192, 256, 1024), (8192, 1024, 256),
                  (65536, 256, 512), (32768, 256, 512), (16384, 512, 256), (65536, 512, 256), (32768, 512, 256)]:
    x = torch.randn(R, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    g = torch.randn(R, N, device=dev, dtype=torch.bfloat16)
    nt = t(lambda: x @ w.t())
    tn = t(lambda: g.t() @ x)
    print(f"R={R:6d} N={N:5d} K={K:5d}: x W^T {nt:6.1f} us ({2*R*N*K/nt/1e6:6.0f} TFLOP/s)   dy^T x {tn:6.1f} us ({2*R*N*K/tn/1e6:6.0f} TFLOP/s)", flush=True)
