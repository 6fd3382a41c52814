"""Scratch: GPU time of torch-level ops (by name and input shapes) in one PN2-MSG training step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointcloud_bridge_amd import rowmlp, parallel
rowmlp.set_precision("bf16")
dev = torch.device("cuda")
torch.manual_seed(42)
name = sys.argv[1] if len(sys.argv) > 1 else "pn2_msg"
model, cdim = bench.build_model(name)
model = model.to(dev).train()
B, N = (8, 8192) if name == "dgcnn" else (16, 16384)
xyz, colors, labels = bench.synthetic_batch(B, N, 1000, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
bucket = parallel.FlatGradAllReduce(model.parameters())
def step():
    bucket.zero()
    loss = bench.loss_fn(model(xyz, colors), labels, cdim)
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", 0)
    if t > 0 and not e.key.startswith("void ") and "anonymous namespace" not in e.key and not e.key.startswith("Cijk"):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
tot = sum(r[0] for r in rows)
print(f"ops with GPU time: {tot/1e3:.2f} ms")
for t, n, k, sh in sorted(rows, reverse=True)[:60]:
    print(f"{t:9.1f} us {n:4d}x  {k:34s} {sh}")
