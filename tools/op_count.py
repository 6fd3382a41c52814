"""Scratch: which torch ops (and from where) launch kernels in one PN2-MSG training step."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointcloud_bridge_amd import rowmlp, parallel
rowmlp.set_precision("bf16")
dev = torch.device("cuda")
torch.manual_seed(42)
model, cdim = bench.build_model(sys.argv[1] if len(sys.argv) > 1 else "pn2_msg")
model = model.to(dev).train()
B, N = (8, 8192) if (len(sys.argv) > 1 and sys.argv[1] == "dgcnn") else (16, 16384)
xyz, colors, labels = bench.synthetic_batch(B, N, 1000, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
bucket = parallel.FlatGradAllReduce(model.parameters())
def step():
    bucket.zero()
    loss = bench.loss_fn(model(xyz, colors), labels, cdim)
    if hasattr(model, "prefetch"):
        model.prefetch(xyz)
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
# kernel launches attributed to the innermost python frame inside this repo
cnt = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        where = "?"
        for fr in (e.stack or []):
            if "pointcloud" in fr or "bench.py" in fr or "parallel.py" in fr:
                where = fr.split("/")[-1]
                break
        cnt[(e.name, where)] += len(e.kernels)
tot = sum(cnt.values())
print("kernel launches attributed:", tot)
for (name, where), c in cnt.most_common(70):
    print(f"{c:5d}  {name:40s} {where}")
