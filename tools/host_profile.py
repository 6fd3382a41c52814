"""Scratch: where the host time of one step goes (cProfile, tottime)."""
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointcloud_bridge_amd import rowmlp, parallel
rowmlp.set_precision("bf16")
torch.manual_seed(42)
name = sys.argv[1] if len(sys.argv) > 1 else "pn2_msg"
model, cdim = bench.build_model(name); model = model.cuda().train()
bucket = parallel.FlatGradAllReduce(model.parameters())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
B, N = (8, 8192) if name == "dgcnn" else (16, 16384)
xyz, colors, labels = bench.synthetic_batch(B, N, 0, "cuda")
def step():
    bucket.zero(); loss = bench.loss_fn(model(xyz, colors), labels, cdim)
    if hasattr(model, "prefetch"): model.prefetch(xyz)
    loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); host = 0.0
for _ in range(10):
    h0 = time.perf_counter(); step(); host += time.perf_counter() - h0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"{name}: host enqueue {host/10*1e3:.2f} ms/step, wall {wall/10*1e3:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
