"""Scratch: how many queries of DGCNN's feature-space graphs the screening pass hands to the exact kernels (bench shape,
a few training steps so that the features are not the initial ones)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointcloud_bridge_amd import ops, rowmlp
torch.manual_seed(0)
model, cdim = bench.build_model("dgcnn")
model = model.cuda().train()
B, N = 8, 8192
xyz, colors, labels = bench.synthetic_batch(B, N, 5, "cuda")
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
with rowmlp.precision("bf16"):
    for step in range(int(os.environ.get("STEPS", "30"))):
        rec = step % 10 == 0 or step == 29
        if rec: ops.collect_knn_stats(True)
        opt.zero_grad(set_to_none=True)
        loss = bench.loss_fn(model(xyz, colors), labels, cdim)
        loss.backward(); opt.step()
        if rec:
            st = ops.collect_knn_stats(False)
            print(f"step {step} loss {float(loss):.3f}: " + "; ".join(f"D={d} recomputed {int(c.sum())} (max/scene {int(c.max())})" for (_, _, d, _, c) in st), flush=True)
