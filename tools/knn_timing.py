"""Scratch: per-segment s_memtime totals of one wave of the screen kernel (build with -DKNN_EXP_TIMING, PCB_LIB=...)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import _lib
from pointcloud_bridge_amd.ops import _launch
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B, N, k = 8, 8192, 20
torch.manual_seed(0)
x = torch.randn(B, N, D, device="cuda")
lib = _lib.load()
ws = torch.zeros(lib.pcb_knn_screen_workspace(B, N, D, k), dtype=torch.uint8, device="cuda")
norms = torch.empty(B, N, device="cuda"); out = torch.empty(B, N, k, dtype=torch.int64, device="cuda")
for _ in range(2):
    _launch("pcb_knn_screened", 0, x.data_ptr(), B, N, D, k, norms.data_ptr(), ws.data_ptr(), out.data_ptr())
torch.cuda.synchronize()
qtk = ws[8 * B + 4 * B * N: 8 * B + 8 * B * N].view(torch.float32)
names = ["tt-steps", "stash+wait", "barrier", "fetch issue", "last step"]
for w in range(4):
    v = qtk[N - 64 + w * 8: N - 64 + w * 8 + 5].tolist()
    print(f"wave {w}: " + "  ".join(f"{n} {c:9.0f}" for n, c in zip(names, v)), " total", sum(v))
