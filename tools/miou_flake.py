"""Scratch: spread of the short-training mIoU used by tests/test_gpu_bf16.py."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import train, rowmlp as rm
from pointcloud_bridge_amd.models.containers import PointNet2
from pointcloud_bridge_amd.models.pointnet2_utils import FeaturePropagation
enc = [(256, 0.2, 16, 6, [32, 32, 64]), (64, 0.4, 16, 67, [64, 64, 128]), (16, 0.8, 16, 131, [128, 128, 256])]
data = train.synthetic_scenes(8, 1024, seed=0, device="cuda")
val = train.synthetic_scenes(4, 1024, seed=1, device="cuda")
for rep in range(6):
    out = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(42)
        model = PointNet2(5, encoder=enc)
        model.fp3 = FeaturePropagation(256 + 128, [128, 128]); model.fp2 = FeaturePropagation(128 + 64, [128, 64]); model.fp1 = FeaturePropagation(64, [128, 128, 128])
        model = model.cuda(); rm.set_precision(mode)
        tr = train.Trainer(model, 5, lr=2e-3); torch.manual_seed(0)
        hist = []
        for i in range(120):
            tr.train_step(data)
            if i in (59, 89, 119): hist.append(round(tr.evaluate([val])["miou"], 3))
        out[mode] = hist
        rm.set_precision("fp32")
    print(rep, out, flush=True)
