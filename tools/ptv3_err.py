"""PointTransformerV3 eval logits against the reference fixture in both precision modes (the numbers written in tests/test_gpu_attention.py)."""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests.helpers import load_golden
from tests.test_gpu_attention import _ptv3
from pointcloud_bridge_amd import rowmlp
g = load_golden("model_ptv3")
xyz, colors = torch.from_numpy(g["xyz"]).cuda(), torch.from_numpy(g["colors"]).cuda()
ref = g["logits_eval"]; scale = np.abs(ref).max()
m = _ptv3(g)
for prec in ("fp32", "bf16"):
    rowmlp.set_precision(prec)
    with torch.no_grad():
        got = m(xyz, colors).float().cpu().numpy()
    print(prec, "max", np.abs(got - ref).max() / scale, "mean", np.abs(got - ref).mean() / np.abs(ref).mean(), "argmax agree", (got.argmax(-1) == ref.argmax(-1)).mean())
