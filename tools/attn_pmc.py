"""Scratch: a few calls of the flash-attention kernel at cfg5's shape (B=8, N=4096, H=2, D=192) for a PMC / timing run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import ops
B, N, H, D = 8, int(os.environ.get("ATTN_N", "4096")), 2, 192
torch.manual_seed(0)
qkv = (torch.randn(B, N, 3 * H * D, device="cuda") * 0.5).to(torch.bfloat16)
for _ in range(3): out = ops.attention(qkv, H)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): out = ops.attention(qkv, H)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 10 * 1e3
print(f"attention B={B} N={N} H={H} D={D}: {us:.0f} us  {4.0 * B * H * N * N * D / us / 1e6:.0f} TFLOP/s")
