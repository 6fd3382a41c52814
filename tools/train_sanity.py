"""Scratch: short synthetic training of the flagship nets in fp32 and bf16 modes (mIoU, loss)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import train, rowmlp as rm
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "pn2_msg"
B, N = (4, 2048)
data = train.synthetic_scenes(B, N, seed=0, device="cuda")
val = train.synthetic_scenes(B, N, seed=1, device="cuda")
for mode in ("fp32", "bf16"):
    torch.manual_seed(42)
    model, _ = bench.build_model(name)
    model = model.cuda()
    rm.set_precision(mode)
    tr = train.Trainer(model, 5, lr=2e-3)
    torch.manual_seed(0)
    t0 = time.time(); losses = []
    for i in range(int(os.environ.get("STEPS", "150"))):
        losses.append(float(tr.train_step(data)))
    m = tr.evaluate([val])
    rm.set_precision("fp32")
    print(name, mode, "loss first/last", round(losses[0], 3), round(sum(losses[-10:]) / 10, 3), "val mIoU", round(m["miou"], 3), "OA", round(m["oa"], 3), f"{time.time()-t0:.1f}s", flush=True)
