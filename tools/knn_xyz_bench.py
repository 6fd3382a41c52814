"""Grid kNN (csrc/knngrid.hip) against the all-pairs kernel on coordinates, at the shapes the models use."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointcloud_bridge_amd import ops
for B, N, k, shape in ((16, 16384, 32, (1, 1, 1)), (16, 16384, 16, (1, 1, 1)), (8, 8192, 20, (1, 1, 1)), (16, 512, 16, (1, 1, 1)),
                       (16, 16384, 32, (1, 0.1, 0.08)), (16, 16384, 32, (1, 1, 0.02))):
    xyz = (bench.synthetic_batch(B, N, 0, "cuda")[0] * torch.tensor(shape, device="cuda")).contiguous()
    res = []
    for grid in (True, False):
        ops.set_grid_knn(grid)
        for _ in range(2): out = ops.knn(xyz, k)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): out = ops.knn(xyz, k)
        b.record(); torch.cuda.synchronize()
        res.append((a.elapsed_time(b) / 5 * 1e3, out))
    ops.set_grid_knn(True)
    print(f"B={B} N={N} k={k} extents {shape}: grid {res[0][0]:.0f} us, all pairs {res[1][0]:.0f} us, equal {torch.equal(res[0][1], res[1][1])}")
