"""Scratch: host enqueue time vs GPU time of one bench step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointcloud_bridge_amd import rowmlp, parallel
rowmlp.set_precision("bf16")
torch.manual_seed(42)
model, cdim = bench.build_model("pn2_msg"); model = model.cuda().train()
bucket = parallel.FlatGradAllReduce(model.parameters())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
xyz, colors, labels = bench.synthetic_batch(16, 16384, int(os.environ.get("SEED", "0")), "cuda")
NSTEP = int(os.environ.get("NSTEP", "10"))
def step(prefetch):
    bucket.zero(); loss = bench.loss_fn(model(xyz, colors), labels, cdim)
    if prefetch: model.prefetch(xyz)
    loss.backward(); opt.step()
for pf in (False, True):
    for _ in range(5): step(pf)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for _ in range(NSTEP):
        h0 = time.perf_counter(); step(pf); host += time.perf_counter() - h0
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    print(f"prefetch={pf}: host enqueue {host/NSTEP*1e3:.2f} ms/step, wall {wall/NSTEP*1e3:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step(False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
