"""Scratch: slowdown of an input-gradient GEMM while the FPS kernel runs on a side stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import _lib, ops
L = _lib.load()
R, K, N = 262144, 256, 264
x = torch.randn(R, K, device="cuda").to(torch.bfloat16); y = torch.randn(R, K, device="cuda").to(torch.bfloat16)
w = torch.randn(N, K, device="cuda").to(torch.bfloat16); out = torch.empty(R, N, device="cuda", dtype=torch.bfloat16)
v = [torch.rand(K, device="cuda") for _ in range(4)]
xyz = torch.rand(int(os.environ.get("FPS_B", "16")), 16384, 3, device="cuda"); start = torch.zeros(xyz.shape[0], dtype=torch.long, device="cuda")
side = torch.cuda.Stream()
def gemm():
    L.pcb_gemm_nt_bf16(2, x.data_ptr(), y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), 0, 0, 1, 1,
                       w.data_ptr(), R, N, K, out.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
def run(with_fps):
    torch.cuda.synchronize()
    if with_fps:
        with torch.cuda.stream(side):
            ops.furthest_point_sample(xyz, 1024, start)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(8): gemm()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 8 * 1e3
for _ in range(3): run(False)
print("gemm alone      %.1f us" % run(False))
print("gemm beside FPS %.1f us" % run(True))
print("gemm alone      %.1f us" % run(False))
print("gemm beside FPS %.1f us" % run(True))
