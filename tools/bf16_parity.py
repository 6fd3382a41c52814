"""Measure the bf16 engine against the REFERENCE fixtures (tests/golden/model_*.npz): the numbers the
bars of tests/test_gpu_round2.py::test_bf16_networks_against_reference_fixtures are set from.
    python tools/bf16_parity.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests.test_gpu_round2 import _bf16_errors  # noqa: E402
from tests.helpers import load_golden  # noqa: E402
from tests.test_gpu_modules import build, dev, grad_norms, run_seg  # noqa: E402

for name, kw in (("model_pn2_ssg", dict(cls="PointNet2", rgb_skip=False)),
                 ("model_pn2_ssg_skip", dict(cls="PointNet2", rgb_skip=True)),
                 ("model_pn2_msg", dict(cls="PointNet2MSG"))):
    print(name, {k: f"{v:.3e}" for k, v in _bf16_errors(name, kw).items()})

from pointcloud_bridge_amd import rowmlp  # noqa: E402
from pointcloud_bridge_amd.models.DGCNN import DGCNN  # noqa: E402
g = load_golden("model_dgcnn")
for prec in ("fp32", "bf16"):
    for k in (20, 8):
        rowmlp.set_precision(prec)
        model = build(DGCNN, g["init_seed"], 5, k=k)
        le, lt, loss = run_seg(model, dev(g["xyz"]), dev(g["colors"]), dev(g["labels"]), int(g["fwd_seed"]), 2)
        rowmlp.set_precision("fp32")
        out = {}
        for tag, got, ref in (("eval", le, g[f"k{k}_logits_eval"]), ("train", lt, g[f"k{k}_logits_train"])):
            d = np.abs(got.float().detach().cpu().numpy() - ref)
            out[f"{tag}_max"] = d.max() / np.abs(ref).max()
            out[f"{tag}_mean"] = d.mean() / np.abs(ref).mean()
            out[f"{tag}_frac1e-4"] = np.mean(d / np.abs(ref).max() < 1e-4)
        out["loss"] = abs(loss - float(g[f"k{k}_loss"])) / abs(float(g[f"k{k}_loss"]))
        gn, ref = grad_norms(model), g[f"k{k}_grad_norms"]
        big = ref > 1e-3 * ref.max()
        r = np.abs(gn[big] - ref[big]) / ref[big]
        out["gn_median"], out["gn_max"] = np.median(r), r.max()
        print("dgcnn", prec, "k", k, {a: f"{float(b):.3e}" for a, b in out.items()})
