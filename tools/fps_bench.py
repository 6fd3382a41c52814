"""Scratch: FPS kernel timing at the bench shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import ops
torch.manual_seed(0)
for B, N, S in ((16, 16384, 1024), (16, 1024, 512), (16, 512, 128), (4, 4096, 1024), (16, 4096, 1024), (16, 3000, 512), (8, 8192, 2048)):
    x = torch.rand(B, N, 3, device="cuda"); st = torch.zeros(B, dtype=torch.long, device="cuda")
    for _ in range(2): ops.furthest_point_sample(x, S, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.furthest_point_sample(x, S, st)
    b.record(); torch.cuda.synchronize()
    print(f"fps B={B} N={N} S={S}: {a.elapsed_time(b)/5*1e3:.0f} us  ({a.elapsed_time(b)/5*1e3/S:.2f} us/iter)")
# anisotropic clouds (bridge-like extents): the cell code adapts its split order to them
for shape in ((1.0, 0.1, 0.08), (1.0, 1.0, 0.05)):
    x = (torch.rand(16, 16384, 3, device="cuda") * torch.tensor(shape, device="cuda")).contiguous()
    st = torch.zeros(16, dtype=torch.long, device="cuda")
    for _ in range(2): ops.furthest_point_sample(x, 1024, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.furthest_point_sample(x, 1024, st)
    b.record(); torch.cuda.synchronize()
    print(f"fps B=16 N=16384 S=1024 extents {shape}: {a.elapsed_time(b)/5*1e3:.0f} us")
