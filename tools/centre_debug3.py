import os, sys
import numpy as np, torch, torch.nn as nn
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from pointcloud_bridge_amd import rowmlp as rm
name = sys.argv[1] if len(sys.argv) > 1 else "dgcnn"
B, N = (2, 1024)
xyz, colors, labels = bench.synthetic_batch(B, N, 5, "cuda")
res = {}
for tag, mode, centring in (("fp32", "fp32", False), ("bf16 plain", "bf16", False), ("bf16 centred", "bf16", True), ("bf16 centred again", "bf16", True)):
    torch.manual_seed(42)
    model, cdim = bench.build_model(name)
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.eval()
    old = rm.set_centring(centring)
    with rm.precision(mode):
        for rep in range(2 if "again" in tag else 1):
            model.zero_grad(set_to_none=True)
            torch.manual_seed(9)
            logits = model(xyz, colors)
            loss = bench.loss_fn(logits, labels, cdim)
            loss.backward()
    rm.set_centring(old)
    res[tag] = (logits.detach().float(), float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
ref = res["fp32"]
for tag in list(res)[1:]:
    l, loss, g = res[tag]
    print(tag, "logits mean rel", float((l - ref[0]).abs().mean() / ref[0].abs().mean()), "loss", loss, ref[1])
    rows = []
    for n in g:
        r = ref[2][n]
        if float(r.norm()) > 1e-3 * max(float(v.norm()) for v in ref[2].values()):
            rows.append((float((g[n] - r).norm() / r.norm()), n))
    rows.sort(reverse=True)
    nr = [(round(float(g[n].norm() / ref[2][n].norm()), 3), n) for n in g if float(ref[2][n].norm()) > 1e-3 * max(float(v.norm()) for v in ref[2].values())]
    print("   norm ratios:", nr)
    print("   worst grads:", [(round(e, 3), n) for e, n in rows[:6]], "median", round(float(np.median([e for e, _ in rows])), 4))
