"""Scratch: one screened and one exact kNN call (D from argv) for a PMC run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import ops
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
x = torch.randn(8, 8192, D, device="cuda")
for flag in (True, False):
    ops.set_screen_knn(flag)
    ops.knn(x, 20)
torch.cuda.synchronize()
