"""mIoU of the short synthetic training (tests/test_gpu_bf16.py) in the reproducible mode: fp32 rows against bf16 rows,
every configuration twice (run-to-run spread must be ZERO), over several model seeds and training lengths.
    python tools/miou_det.py [steps ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import ops, train, rowmlp as rm
from pointcloud_bridge_amd.models.containers import PointNet2
from pointcloud_bridge_amd.models.pointnet2_utils import FeaturePropagation
enc = [(256, 0.2, 16, 6, [32, 32, 64]), (64, 0.4, 16, 67, [64, 64, 128]), (16, 0.8, 16, 131, [128, 128, 256])]
data = train.synthetic_scenes(8, 1024, seed=0, device="cuda")
val = train.synthetic_scenes(4, 1024, seed=1, device="cuda")
steps_list = [int(a) for a in sys.argv[1:]] or [120, 240]
ops.set_deterministic(True)

def run(mode, seed, steps):
    torch.manual_seed(seed)
    model = PointNet2(5, encoder=enc)
    model.fp3 = FeaturePropagation(256 + 128, [128, 128]); model.fp2 = FeaturePropagation(128 + 64, [128, 64]); model.fp1 = FeaturePropagation(64, [128, 128, 128])
    model = model.cuda()
    with rm.precision(mode):
        tr = train.Trainer(model, 5, lr=2e-3)
        torch.manual_seed(0); torch.cuda.manual_seed(0)
        hist = {}
        for i in range(max(steps_list)):
            tr.train_step(data)
            if i + 1 in steps_list:
                hist[i + 1] = tr.evaluate([val])["miou"]
    return hist

for seed in (42, 43, 44, 45, 46, 47):
    res = {}
    for mode in ("fp32", "bf16"):
        a, b = run(mode, seed, None), run(mode, seed, None)
        res[mode] = a
        assert a == b, (mode, a, b)
    print("seed", seed, {s: (round(res["fp32"][s], 4), round(res["bf16"][s], 4), round(res["bf16"][s] - res["fp32"][s], 4)) for s in steps_list}, "(fp32, bf16, diff); repeat runs identical", flush=True)
