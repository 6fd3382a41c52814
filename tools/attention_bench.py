"""pcb_attention_fwd_bf16 against torch's bf16 scaled_dot_product_attention at the PTv3 shapes (head_dim 192, 2 heads)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import ops

def timeit(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

H, D = 2, 192
for B, N in [(8, 4096), (4, 8192), (2, 16384), (1, 32768), (1, 65536)]:
    qkv = (torch.randn(B, N, 3 * H * D, device="cuda") * 0.5).to(torch.bfloat16)
    ms = timeit(lambda: ops.attention(qkv, H))
    fl = 4.0 * B * H * N * N * D
    q, k, v = qkv.reshape(B, N, 3, H, D).permute(2, 0, 3, 1, 4).unbind(0)
    try:
        ms_t = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
    except Exception as e:
        ms_t = float("nan")
    print(f"B={B} N={N}: own {ms:8.3f} ms = {fl / ms / 1e9:7.1f} TFLOP/s | torch SDPA (permuted views) {ms_t:8.3f} ms = {fl / ms_t / 1e9:7.1f} TFLOP/s")
