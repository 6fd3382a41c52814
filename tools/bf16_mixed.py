"""Where must precision go to bring the bf16 engine's TRAIN-mode logits close to the reference?

Two experiments on the reference fixtures (tests/golden/model_pn2_*.npz):

  1. noise amplification: the fp32 engine with ONE relative perturbation of bf16 size (2^-9, uniform) injected into the
     output of one stage -- how large is it at the logits?  (A freshly initialised ReLU + BatchNorm network amplifies
     independent noise relative to the signal by ~1.2 per layer: ReLU halves the noise variance but turns 2/3 of the
     signal's variance into a mean that the next BatchNorm removes.)
  2. mixed precision by stage (rowmlp.bind_precision): stages run in fp32 rows from the input up to a cut, bf16 rows after
     it, with and without centred storage of the pre-BatchNorm rows.

    python tools/bf16_mixed.py [pn2_msg|pn2_ssg]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloud_bridge_amd import rowmlp  # noqa: E402
from pointcloud_bridge_amd.models import containers  # noqa: E402
from tests.helpers import load_golden  # noqa: E402
from tests.test_gpu_modules import build, dev, grad_norms, run_seg  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pn2_msg"
g = load_golden("model_" + name)
cls = containers.PointNet2 if name == "pn2_ssg" else containers.PointNet2MSG
stages = ["sa1", "sa2", "sa3", "fp3", "fp2", "fp1"]
xyz, colors, labels = dev(g["xyz"]), dev(g["colors"]), dev(g["labels"])


def errors(model):
    le, lt, loss = run_seg(model, xyz, colors, labels, int(g["fwd_seed"]), 1)
    out = {}
    for tag, got, ref in (("eval", le, g["logits_eval"]), ("train", lt, g["logits_train"])):
        d = np.abs(got.float().detach().cpu().numpy() - ref)
        out[f"{tag}_max"] = float(d.max() / np.abs(ref).max())
        out[f"{tag}_mean"] = float(d.mean() / np.abs(ref).mean())
    gn, ref = grad_norms(model), g["grad_norms"]
    big = ref > 1e-3 * ref.max()
    r = np.abs(gn[big] - ref[big]) / ref[big]
    out["gn_median"], out["gn_max"] = float(np.median(r)), float(r.max())
    return out


def fmt(e):
    return "  ".join(f"{k} {v:.2e}" for k, v in e.items())


print("== 1. one perturbation of relative size 2^-9 (uniform in +-2^-9) at a stage's output, fp32 engine")
for s in stages:
    model = build(cls, g["init_seed"], 5)
    gen = torch.Generator(device="cuda").manual_seed(1)

    def hook(mod, args, out, gen=gen):
        def noisy(t):
            return t * (1 + (torch.rand(t.shape, device=t.device, generator=gen) * 2 - 1) * 2.0 ** -9)
        return (out[0], noisy(out[1])) if isinstance(out, tuple) else noisy(out)

    h = getattr(model, s).register_forward_hook(hook)
    e = errors(model)
    h.remove()
    print(f"  noise behind {s:4s}: train_max {e['train_max']:.2e} train_mean {e['train_mean']:.2e}   (eval_max {e['eval_max']:.2e})")

print("== 2. fp32 rows up to a cut, bf16 rows behind it")
for centring in (False, True):
    old = rowmlp.set_centring(centring)
    for cut in range(len(stages) + 1):
        model = build(cls, g["init_seed"], 5)
        for s in stages[:cut]:
            rowmlp.bind_precision(getattr(model, s), "fp32")
        with rowmlp.precision("bf16"):
            e = errors(model)
        print(f"  centring {int(centring)}  fp32: {','.join(stages[:cut]) or '-':24s} {fmt(e)}")
    rowmlp.set_centring(old)
