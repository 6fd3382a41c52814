"""Which allocations made DURING the capture of a DGCNN step did NOT come from the graph's private
pool?  (A graph replays kernels with baked addresses; a buffer from the ordinary pool is handed out
again by the allocator after the capture, and whatever is written there feeds the next replay.)
Pure host-side diagnostic: captures, snapshots the allocator history, replays nothing."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloud_bridge_amd import rowmlp  # noqa: E402
from pointcloud_bridge_amd.models.DGCNN import DGCNN  # noqa: E402
from tests.helpers import load_golden  # noqa: E402
from tests.test_gpu_modules import build, dev  # noqa: E402

g = load_golden("model_dgcnn")
xyz, colors, labels = dev(g["xyz"]), dev(g["colors"]), dev(g["labels"])
model = build(DGCNN, g["init_seed"], 5, k=20).train()
rowmlp.set_precision("bf16")


def step():
    F.cross_entropy(model(xyz, colors).reshape(-1, 5), labels.reshape(-1)).backward()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for p in model.parameters():
        p.grad = None
    step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for p in model.parameters():
    p.grad = None
torch.cuda.memory._record_memory_history(enabled="all", context="all", stacks="python", max_entries=200000)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
private = {s["address"]: s for s in snap["segments"] if tuple(s.get("segment_pool_id", (0, 0))) != (0, 0)}
print("segments:", len(snap["segments"]), "private:", len(private))
ranges = sorted((s["address"], s["address"] + s["total_size"]) for s in private.values())


def in_private(addr):
    return any(lo <= addr < hi for lo, hi in ranges)


outside = 0
for trace in snap["device_traces"]:
    for ev in trace:
        if ev["action"] == "alloc" and not in_private(ev["addr"]):
            outside += 1
            frames = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", [])
                      if "site-packages" not in f["filename"] and "dist-packages" not in f["filename"]][:6]
            print(f"ALLOC outside the private pool: {ev['size']} bytes  stream {ev.get('stream')}  {' <- '.join(frames)}")
print("allocations outside the private pool during capture:", outside)
