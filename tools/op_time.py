"""GPU time of the torch-level (ATen) ops of one bench training step, by op name and input shapes, with the
autograd node that issued them in the backward pass.  Library kernels are skipped: what is listed here is the
glue a fused kernel or a layout change could remove.

    python tools/op_time.py [model]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("model", nargs="?", default="pn2_msg")
a = ap.parse_args()
args = argparse.Namespace(no_dropout=False, no_prefetch=False, dump=False)
dev = torch.device("cuda", 0)
B, N = (8, 8192) if a.model in ("dgcnn", "bridgeseg") else (16, 16384)
run = bench.Run(args, a.model, "bf16", B, N, 0, 1, dev)
for _ in range(4):
    run.train_step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    run.train_step()
    torch.cuda.synchronize()
events = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU), key=lambda e: e.time_range.start)
nodes = [e for e in events if e.name.startswith("autograd::engine::evaluate_function") or e.name.endswith("Backward")]
rows = collections.defaultdict(lambda: [0.0, 0])
for e in events:
    t = getattr(e, "self_device_time_total", 0)
    if t <= 0 or not e.name.startswith("aten::"):
        continue
    owner = ""
    for n in nodes:
        if n.time_range.start <= e.time_range.start and e.time_range.end <= n.time_range.end and n.thread == e.thread:
            owner = n.name.replace("autograd::engine::evaluate_function: ", "")
    key = (e.name, str(e.input_shapes)[:90], owner[:40])
    rows[key][0] += t
    rows[key][1] += 1
tot = sum(v[0] for v in rows.values())
print(f"ATen ops with GPU time: {tot / 1e3:.2f} ms in {sum(v[1] for v in rows.values())} calls")
for (name, shapes, owner), (t, n) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{t:8.1f} us {n:3d}x {name:28s} {owner:40s} {shapes}")
