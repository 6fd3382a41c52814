"""Golden vectors for the bridge encoders from the REFERENCE (CPU, fp32).

Run in the build container only (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bridge.py

Reference modules imported: models/attention_modules.py (BridgeStructureEncoding,
GeometricFeatureExtraction, ColorFeatureExtraction, CompositeFeatureFusion) and models/model.py
(EnhancedPointNet2).  Fixtures hold inputs, seeds and the reference's outputs only.  Parameters are
not stored: the build's modules create theirs in the reference's order, so the same seed gives the
same tensors -- asserted key by key before anything is written.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_golden import unit_ball_cloud  # noqa: E402
from make_golden_modules import FWD_SEED, INIT_SEED, _run_seg, _same_params  # noqa: E402


def _pair(ref_ctor, my_ctor):
    torch.manual_seed(INIT_SEED)
    ref = ref_ctor()
    torch.manual_seed(INIT_SEED)
    _same_params(ref, my_ctor())
    return ref


def make_encoders():
    from models import attention_modules as ram
    from pointcloud_bridge_amd.models import attention_modules as mam

    g = torch.Generator().manual_seed(19)
    B, N = 2, 1024
    xyz = unit_ball_cloud(g, B, N)
    res = {"xyz": xyz.numpy(), "init_seed": np.int64(INIT_SEED)}

    enc = _pair(lambda: ram.BridgeStructureEncoding(3, 32, 4), lambda: mam.BridgeStructureEncoding(3, 32, 4))
    for k in (16, 32):
        idx = torch.cdist(xyz, xyz).topk(k, dim=-1, largest=False)[1]        # attention_modules.py:584-586
        flat = (idx + torch.arange(B).view(-1, 1, 1) * N).view(-1)
        rel = xyz.view(B * N, 3)[flat].view(B, N, k, 3) - xyz.unsqueeze(2)   # :595-600
        res[f"idx{k}"] = idx.numpy().astype(np.int32)
        res[f"desc{k}"] = enc.get_structure_features(rel).numpy()            # :620
    res["abs_enc"] = enc.compute_absolute_position_encoding(xyz).numpy()

    def fwd_bwd(tag, mod, *inputs):
        for mode in ("eval", "train"):
            mod.train(mode == "train")
            out = mod(*inputs)
            res[f"{tag}_{mode}"] = out.detach().numpy()
            if mode == "train":
                mod.zero_grad()
                (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum().backward()
                res[f"{tag}_grad_norms"] = np.array([float(p.grad.norm()) for p in mod.parameters()])
                for i in inputs:
                    if i.grad is not None:
                        res[f"{tag}_grad_in"] = i.grad.numpy()

    fwd_bwd("enc", enc, xyz)
    feats = torch.randn(B, 32, N, generator=g)
    res["geo_x"] = feats.numpy()
    geo = _pair(lambda: ram.GeometricFeatureExtraction(32), lambda: mam.GeometricFeatureExtraction(32))
    fwd_bwd("geo", geo, feats.clone().requires_grad_(True), xyz)
    colors = torch.rand(B, 3, N, generator=g)
    res["colors"] = colors.numpy()
    col = _pair(lambda: ram.ColorFeatureExtraction(3, 6), lambda: mam.ColorFeatureExtraction(3, 6))
    fwd_bwd("col", col, colors, xyz)
    fus = _pair(lambda: ram.CompositeFeatureFusion(3, 6), lambda: mam.CompositeFeatureFusion(3, 6))
    sp = torch.randn(B, 3, N, generator=g)
    cf = torch.randn(B, 6, N, generator=g)
    res.update(fus_s=sp.numpy(), fus_c=cf.numpy())
    fwd_bwd("fus", fus, sp, cf)
    np.savez_compressed(os.path.join(HERE, "bridge_encoders.npz"), **res)
    print("bridge_encoders.npz", len(res), "arrays")


def make_model():
    from models.model import EnhancedPointNet2 as RefNet
    from pointcloud_bridge_amd.models.containers import EnhancedPointNet2

    g = torch.Generator().manual_seed(23)
    B, N = 2, 2048
    xyz = unit_ball_cloud(g, B, N)
    colors = torch.rand(B, N, 3, generator=g)
    labels = torch.randint(0, 5, (B, N), generator=g)
    out = {"xyz": xyz.numpy(), "colors": colors.numpy(), "labels": labels.numpy(),
           "init_seed": np.int64(INIT_SEED), "fwd_seed": np.int64(FWD_SEED)}
    ref = _pair(lambda: RefNet(5), lambda: EnhancedPointNet2(5))
    out.update(_run_seg(ref, (xyz, colors), labels, 1))
    # The same network in fp64: how far the fp32 outputs above are from exact arithmetic (this deep,
    # with batch statistics over as few as 256 rows: 2e-4 on the train-mode logits, percents on the
    # first layers' gradient norms) -- the accuracy of the fp32 vectors as a yardstick.  Evaluated
    # with the CPU port (oracle/torch_port.py, which tests/test_torch_port_cpu.py pins to these very
    # fp32 vectors at 1e-5): the reference's own farthest_point_sample hard-codes an fp32 buffer
    # (pointnet2_utils.py:68,77) and raises on fp64 input.
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import torch_port as port

    class _Port64(torch.nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, xyz, colors):
            return port.run(self.net, xyz, colors)

    torch.manual_seed(INIT_SEED)
    p64 = _Port64(EnhancedPointNet2(5).double())
    r64 = _run_seg(p64, (xyz.double(), colors.double()), labels, 1)
    out.update({k + "_f64": v for k, v in r64.items()})
    np.savez_compressed(os.path.join(HERE, "model_bridgeseg.npz"), **out)
    print("model_bridgeseg", out["loss"], out["logits_train"].shape)


def make_loss():
    """BridgeStructureLoss (models/model.py:169-260) on six small batches that reach every branch:
    all classes present, a class absent from the labels, a class never predicted, everything
    predicted 'other', labels all 'other', one class dominating a scene."""
    from models.model import BridgeStructureLoss as RefLoss

    g = torch.Generator().manual_seed(1)
    res = {}
    for case in range(6):
        B, N = 3, 700
        out = torch.randn(B, 5, N, generator=g) * 2
        pts = torch.randn(B, N, 3, generator=g)
        lab = torch.randint(0, 5, (B, N), generator=g)
        if case == 1:
            lab[lab == 3] = 0
        if case == 2:
            lab[lab == 1] = 2
            out[:, 4] -= 100
        if case == 3:
            out[:, 1:] -= 100
        if case == 4:
            lab[:] = 0
        if case == 5:
            out[0, 2] += 100
            out[1, 4] += 100
        res[f"c{case}_outputs"], res[f"c{case}_points"], res[f"c{case}_labels"] = out.numpy(), pts.numpy(), lab.numpy()
        for tag, (alpha, margin) in (("a80", (80, 0.3)), ("a20", (20.0, 0.2))):   # trainer's / default
            o = out.clone().requires_grad_(True)
            loss = RefLoss(alpha=alpha, rel_margin=margin)(o, lab, pts)
            loss.backward()
            res[f"c{case}_{tag}_loss"] = np.float64(loss.item())
            res[f"c{case}_{tag}_grad"] = o.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "bridge_loss.npz"), **res)
    print("bridge_loss.npz", len(res), "arrays")


if __name__ == "__main__":
    if "loss" in sys.argv[1:]:
        make_loss()
        sys.exit(0)
    if "model" not in sys.argv[1:]:
        make_encoders()
        make_loss()
    make_model()
