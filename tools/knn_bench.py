"""Scratch: kNN kernel timing at the DGCNN bench shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import ops
torch.manual_seed(0)
for B in (int(v) for v in os.environ.get("KNN_B", "8").split(",")):
    for D in (3, 64):
        x = torch.randn(B, 8192, D, device="cuda")
        for _ in range(3): ops.knn(x, 20)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.knn(x, 20)
        b.record(); torch.cuda.synchronize()
        print(f"knn B={B} D={D}: {a.elapsed_time(b)/10*1e3:.0f} us")
