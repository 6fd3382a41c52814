"""Which parameter gradients differ between two identical reproducible-mode steps?  python tools/det_debug.py [model] [precision]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from pointcloud_bridge_amd import ops, rowmlp
name = sys.argv[1] if len(sys.argv) > 1 else "pn2_ssg"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
ops.set_deterministic(True)

def run():
    torch.manual_seed(42)
    model, cdim = bench.build_model(name)
    model = model.cuda().train()
    B, N = (2, 2048) if name != "dgcnn" else (2, 1024)
    xyz, colors, labels = bench.synthetic_batch(B, N, 5, "cuda")
    torch.manual_seed(9); torch.cuda.manual_seed(9)
    acts = {}
    hooks = [m.register_forward_hook(lambda mod, a, o, n=n: acts.__setitem__(n, (o[1] if isinstance(o, tuple) else o).detach().float().clone()))
             for n, m in model.named_children()]
    with rowmlp.precision(prec):
        logits = model(xyz, colors)
        loss = bench.loss_fn(logits, labels, cdim)
        loss.backward()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, acts, logits.detach().clone()

a, aa, la = run()
b, ab, lb = run()
print("logits equal", torch.equal(la, lb))
for n in aa:
    if not torch.equal(aa[n], ab[n]):
        print("forward differs at", n, float((aa[n] - ab[n]).abs().max()))
bad = [n for n in a if not torch.equal(a[n], b[n])]
print(len(bad), "of", len(a), "parameter gradients differ")
for n in bad:
    d = (a[n] - b[n]).abs().reshape(a[n].shape[0], -1)
    print("  ", n, float(d.max() / a[n].abs().max()), "coordinate columns", float(d[:, :3].max()), "feature columns", float(d[:, 3:].max()),
          "rows that differ", int((d.max(1)[0] > 0).sum()), "of", d.shape[0])
