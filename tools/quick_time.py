"""Scratch timing of the individual HIP operators at BASELINE sizes (not part of the product)."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pointcloud_bridge_amd import ops

def timeit(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

torch.manual_seed(0)
B, N = 16, 16384
v = torch.randn(B, N, 3); xyz = (v / v.norm(dim=-1, keepdim=True) * torch.rand(B, N, 1) ** (1/3)).cuda()
start = torch.randint(0, N, (B,)).cuda()
for (n, s) in ((16384, 1024), (1024, 512), (512, 128), (4096, 1024), (8192, 2048)):
    x = xyz[:, :n].contiguous()
    print(f"fps {n}->{s}: {timeit(lambda: ops.furthest_point_sample(x, s, start % n)):.3f} ms")
fps = ops.furthest_point_sample(xyz, 1024, start); new_xyz = ops.gather_rows(xyz, fps)
print(f"ball r.1 ns16: {timeit(lambda: ops.ball_query(0.1, 16, xyz, new_xyz)):.3f} ms")
print(f"ball r.2 ns32: {timeit(lambda: ops.ball_query(0.2, 32, xyz, new_xyz)):.3f} ms")
print(f"ball2: {timeit(lambda: ops.ball_query2([0.1,0.2],[16,32], xyz, new_xyz)):.3f} ms")
print(f"three_nn 16384<-1024 k3: {timeit(lambda: ops.three_nn(xyz, new_xyz, 3)):.3f} ms")
print(f"three_nn k4: {timeit(lambda: ops.three_nn(xyz, new_xyz, 4)):.3f} ms")
idx = ops.ball_query(0.2, 32, xyz, new_xyz)
feat = torch.randn(B, N, 3).cuda()
print(f"group C=3: {timeit(lambda: ops.group_points(xyz, new_xyz, feat, idx)):.3f} ms")
f256 = torch.randn(B, 1024, 256).cuda(); x1 = new_xyz; fps2 = ops.furthest_point_sample(x1, 512, start % 1024); nx2 = ops.gather_rows(x1, fps2)
idx2 = ops.ball_query(0.4, 32, x1, nx2)
print(f"group C=256 S=512: {timeit(lambda: ops.group_points(x1, nx2, f256, idx2)):.3f} ms")
d, i = ops.three_nn(xyz, new_xyz, 3)
f128 = torch.randn(B, 1024, 256).cuda()
print(f"interp C=256: {timeit(lambda: ops.three_interpolate(f128, d, i)):.3f} ms")
xk = xyz[:8, :8192].contiguous()
print(f"knn D=3 N=8192 B=8 k=20: {timeit(lambda: ops.knn(xk, 20), n=3, w=1):.3f} ms")
x64 = torch.randn(8, 8192, 64).cuda()
print(f"knn D=64: {timeit(lambda: ops.knn(x64, 20), n=3, w=1):.3f} ms")
idk = ops.knn(x64, 20)
print(f"edge feat D=64: {timeit(lambda: ops.edge_features(x64, idk)):.3f} ms")
