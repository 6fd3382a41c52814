"""Where does the bf16 engine's train-mode logit error come from?  Per top-level stage: relative
error of the bf16 output against the fp32-mode output (same sampling), in train and in eval mode;
per BatchNorm layer: the largest |mean| / std of its input over the channels (recovered from the
running statistics after one step) -- a channel stored in bf16 loses its signal when that ratio
approaches 2^8.
    python tools/bf16_stage_err.py [pn2_ssg|pn2_msg]
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloud_bridge_amd import rowmlp  # noqa: E402
from pointcloud_bridge_amd.models import containers  # noqa: E402
from tests.helpers import load_golden  # noqa: E402
from tests.test_gpu_modules import build, dev, dropout_eval  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pn2_ssg"
g = load_golden("model_" + name)
cls = containers.PointNet2 if name == "pn2_ssg" else containers.PointNet2MSG
xyz, colors = dev(g["xyz"]), dev(g["colors"])
stages = ["sa1", "sa2", "sa3", "fp3", "fp2", "fp1"]


def run(prec, train):
    model = build(cls, g["init_seed"], 5)
    model.train(train)
    dropout_eval(model)
    outs = {}
    hooks = [getattr(model, s).register_forward_hook(lambda m, i, o, s=s: outs.__setitem__(s, (o[1] if isinstance(o, tuple) else o).float().detach()))
             for s in stages]
    rowmlp.set_precision(prec)
    torch.manual_seed(int(g["fwd_seed"]))
    with torch.set_grad_enabled(train):
        outs["logits"] = model(xyz, colors).float().detach()
    rowmlp.set_precision("fp32")
    for h in hooks:
        h.remove()
    return model, outs


for train in (False, True):
    m32, o32 = run("fp32", train)
    m16, o16 = run("bf16", train)
    print("train" if train else "eval")
    for s in stages + ["logits"]:
        a, b = o16[s], o32[s]
        print(f"  {s:7s} max {float((a - b).abs().max() / b.abs().max()):.3e}  mean {float((a - b).abs().mean() / b.abs().mean()):.3e}")
    if train:
        print("  per BatchNorm: largest |mean y| / std y over channels (fp32 run), #channels with ratio > 16")
        for n, mod in m32.named_modules():
            if isinstance(mod, (nn.BatchNorm1d, nn.BatchNorm2d)):
                mean = mod.running_mean / 0.1
                var = ((mod.running_var - 0.9) / 0.1).clamp_min(1e-12)
                ratio = (mean.abs() / var.sqrt())
                print(f"    {n:28s} max ratio {float(ratio.max()):9.2f}   >16: {int((ratio > 16).sum()):4d} of {ratio.numel()}   min std {float(var.sqrt().min()):.3e}")
