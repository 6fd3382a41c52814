"""Scratch: kNN kernel timing at the DGCNN bench shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import ops
torch.manual_seed(0)
for D in (3, 64):
    x = torch.randn(8, 8192, D, device="cuda")
    for _ in range(3): ops.knn(x, 20)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.knn(x, 20)
    b.record(); torch.cuda.synchronize()
    print(f"knn D={D}: {a.elapsed_time(b)/10*1e3:.0f} us")
