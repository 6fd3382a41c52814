"""List the ATen operators and kernels of one eager DGCNN training step (which op launches the
scatter/gather kernels, with what sizes)."""
import os
import sys

import torch
import torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile

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
    for p in model.parameters():
        p.grad = None
    F.cross_entropy(model(xyz, colors).reshape(-1, 5), labels.reshape(-1)).backward()


step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and ("scatter" in e.name or "gather" in e.name or "index" in e.name):
        print("KERNEL", e.name[:110])
for e in prof.key_averages(group_by_input_shape=True):
    if any(s in e.key for s in ("scatter", "gather", "index", "max", "take", "put")):
        print("OP", e.key, e.count, str(e.input_shapes)[:160])
