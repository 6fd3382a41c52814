"""Module- and model-level golden vectors from the REFERENCE (CPU, fp32).  Driven by make_golden.py.

For every case the fixture stores: the inputs, the seeds (parameter init seed, CPU-generator seed
in front of the forward that feeds farthest_point_sample's torch.randint), the reference's outputs
in eval mode and in train mode (Dropout layers kept in eval so no RNG other than FPS is consumed),
the CrossEntropy loss, and gradients (w.r.t. an input where one exists, plus the L2 norm of every
parameter gradient).  Parameters are NOT stored: the build's modules create theirs in the same
order as the reference's, so the same seed yields the same tensors -- asserted here, key by key.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Highway_bridge"
sys.dont_write_bytecode = True
for p in (REF, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from make_golden import unit_ball_cloud  # noqa: E402

INIT_SEED = 42   # tools/debug_module.py:8 of the reference seeds its smoke run with 42
FWD_SEED = 123


def _same_params(ref, mine):
    a, b = ref.state_dict(), mine.state_dict()
    assert list(a.keys()) == list(b.keys()), (list(a.keys())[:5], list(b.keys())[:5])
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k


def _dropout_eval(m):
    for s in m.modules():
        if isinstance(s, nn.Dropout):
            s.eval()


def _grad_norms(m):
    return np.array([0.0 if p.grad is None else float(p.grad.norm()) for _, p in m.named_parameters()],
                    dtype=np.float64)


def _run_seg(model, args, labels, channel_dim):
    """eval logits, train logits (fresh BN stats), loss, param-grad norms."""
    out = {}
    model.eval()
    torch.manual_seed(FWD_SEED)
    with torch.no_grad():
        out["logits_eval"] = model(*args).numpy()
    model.train()
    _dropout_eval(model)
    torch.manual_seed(FWD_SEED)
    logits = model(*args)
    out["logits_train"] = logits.detach().numpy()
    lg = logits if channel_dim == 1 else logits.reshape(-1, logits.shape[-1])
    lb = labels if channel_dim == 1 else labels.reshape(-1)
    loss = F.cross_entropy(lg, lb)
    loss.backward()
    out["loss"] = np.float64(loss.item())
    out["grad_norms"] = _grad_norms(model)
    return out


def make_modules():
    from models import pointnet2_utils as rpu
    from pointcloud_bridge_amd.models import pointnet2_utils as mpu

    g = torch.Generator().manual_seed(7)
    B, N = 2, 512
    xyz = unit_ball_cloud(g, B, N)
    feats = torch.randn(B, 5, N, generator=g)

    def build(cls_name, *a):
        torch.manual_seed(INIT_SEED)
        ref = getattr(rpu, cls_name)(*a)
        torch.manual_seed(INIT_SEED)
        mine = getattr(mpu, cls_name)(*a)
        _same_params(ref, mine)
        return ref

    res = {"xyz": xyz.numpy(), "feats": feats.numpy(), "init_seed": np.int64(INIT_SEED),
           "fwd_seed": np.int64(FWD_SEED)}

    def run_sa(tag, mod):
        for mode in ("eval", "train"):
            mod.train(mode == "train")
            f = feats.clone().requires_grad_(True)
            torch.manual_seed(FWD_SEED)
            new_xyz, out = mod(xyz, f)
            res[f"{tag}_{mode}_new_xyz"] = new_xyz.detach().numpy()
            res[f"{tag}_{mode}_out"] = out.detach().numpy()
            if mode == "train":
                mod.zero_grad()
                (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum().backward()
                res[f"{tag}_grad_feats"] = f.grad.numpy()
                res[f"{tag}_grad_norms"] = _grad_norms(mod)

    run_sa("sa", build("SetAbstraction", 128, 0.3, 16, 8, [16, 16, 32]))
    run_sa("msg", build("MultiScaleSetAbstraction", 128, [0.2, 0.4], [8, 16], 8, [16, 16, 32]))
    # SetAbstraction without features (points=None)
    sa0 = build("SetAbstraction", 64, 0.4, 8, 3, [8, 16])
    sa0.eval()
    torch.manual_seed(FWD_SEED)
    nx, o = sa0(xyz, None)
    res["sa0_eval_out"] = o.detach().numpy()

    # feature propagation: N points <- S centroids
    S = 96
    xyz2 = xyz[:, :S].contiguous()
    p1 = torch.randn(B, 6, N, generator=g)
    p2 = torch.randn(B, 12, S, generator=g)
    res.update(fp_xyz2=xyz2.numpy(), fp_p1=p1.numpy(), fp_p2=p2.numpy())

    def run_fp(tag, mod, points1):
        for mode in ("eval", "train"):
            mod.train(mode == "train")
            a = None if points1 is None else points1.clone().requires_grad_(True)
            c = p2.clone().requires_grad_(True)
            out = mod(xyz, xyz2, a, c)
            res[f"{tag}_{mode}_out"] = out.detach().numpy()
            if mode == "train":
                mod.zero_grad()
                (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum().backward()
                res[f"{tag}_grad_p2"] = c.grad.numpy()
                if a is not None:
                    res[f"{tag}_grad_p1"] = a.grad.numpy()
                res[f"{tag}_grad_norms"] = _grad_norms(mod)

    run_fp("fp", build("FeaturePropagation", 18, [16, 8]), p1)
    run_fp("fp_nop1", build("FeaturePropagation", 12, [16]), None)
    run_fp("efp", build("EnhancedFeaturePropagation", 18, [16, 8]), p1)
    run_fp("efp_skip", build("EnhancedFeaturePropagation", 18, [16, 18]), p1)
    # S == 1: the reference's branch (:181-182) builds a [B,D,N] tensor where [B,N,D] is needed and
    # fails in torch.cat / the first conv for every input; record that it raises.
    fp1 = build("FeaturePropagation", 18, [8])
    fp1.eval()
    try:
        fp1(xyz, xyz2[:, :1].contiguous(), p1, p2[:, :, :1].contiguous())
        res["fp_s1_raises"] = np.int64(0)
    except RuntimeError:
        res["fp_s1_raises"] = np.int64(1)

    np.savez_compressed(os.path.join(HERE, "modules.npz"), **res)
    print("modules.npz", len(res), "arrays")


class _RefMsgTrunk(nn.Module):
    """The SA/FP trunk of the reference's EnhancedPointNet2 (models/model.py:73-99), assembled from
    the REFERENCE's own classes in the order the build's PointNet2MSG creates its sub-modules."""

    def __init__(self, num_classes=5):
        super().__init__()
        from models import pointnet2_utils as rpu
        from models.model import MultiScaleFeatureFusion
        self.sa1 = rpu.MultiScaleSetAbstraction(1024, [0.1, 0.2], [16, 32], 6, [64, 64, 128])
        self.sa2 = rpu.MultiScaleSetAbstraction(512, [0.2, 0.4], [16, 32], 259, [128, 128, 256])
        self.sa3 = rpu.MultiScaleSetAbstraction(128, [0.4, 0.8], [16, 32], 515, [256, 256, 512])
        self.fp3 = rpu.EnhancedFeaturePropagation(1536, [1024, 256])
        self.fp2 = rpu.EnhancedFeaturePropagation(512, [256, 256])
        self.fp1 = rpu.EnhancedFeaturePropagation(256 + 3, [256, 128])
        self.fusion = MultiScaleFeatureFusion([256, 256, 128], 128)
        self.final_fusion = nn.Sequential(nn.Conv1d(384, 128, 1), nn.BatchNorm1d(128), nn.ReLU(),
                                          nn.Dropout(0.5), nn.Conv1d(128, num_classes, 1))

    def forward(self, xyz, features):
        feats = features.transpose(1, 2)
        l1_xyz, l1 = self.sa1(xyz, feats)
        l2_xyz, l2 = self.sa2(l1_xyz, l1)
        l3_xyz, l3 = self.sa3(l2_xyz, l2)
        l2 = self.fp3(l2_xyz, l3_xyz, l2, l3)
        l1 = self.fp2(l1_xyz, l2_xyz, l1, l2)
        l0 = self.fp1(xyz, l1_xyz, feats, l1)
        return self.final_fusion(self.fusion([l2, l1, l0]))


def make_models():
    from models.model import PointNet2 as RefSSG           # models/model.py:12
    from models.pointnet2 import PointNet2 as RefSSGSkip    # models/pointnet2.py:10
    from models.DGCNN import DGCNN as RefDGCNN
    from pointcloud_bridge_amd.models.containers import PointNet2, PointNet2MSG
    from pointcloud_bridge_amd.models.DGCNN import DGCNN

    g = torch.Generator().manual_seed(11)
    B, N = 2, 2048
    xyz = unit_ball_cloud(g, B, N)
    colors = torch.rand(B, N, 3, generator=g)
    labels = torch.randint(0, 5, (B, N), generator=g)
    common = {"xyz": xyz.numpy(), "colors": colors.numpy(), "labels": labels.numpy(),
              "init_seed": np.int64(INIT_SEED), "fwd_seed": np.int64(FWD_SEED)}

    def pair(ref_ctor, my_ctor):
        torch.manual_seed(INIT_SEED)
        ref = ref_ctor()
        torch.manual_seed(INIT_SEED)
        mine = my_ctor()
        _same_params(ref, mine)
        return ref

    cases = {
        "model_pn2_ssg": (pair(lambda: RefSSG(5), lambda: PointNet2(5)), 1),
        "model_pn2_ssg_skip": (pair(lambda: RefSSGSkip(5), lambda: PointNet2(5, rgb_skip=True)), 1),
        "model_pn2_msg": (pair(lambda: _RefMsgTrunk(5), lambda: PointNet2MSG(5)), 1),
    }
    for name, (ref, cdim) in cases.items():
        out = dict(common)
        out.update(_run_seg(ref, (xyz, colors), labels, cdim))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, out["loss"], out["logits_train"].shape)

    # DGCNN: smaller cloud (the reference builds a [B,N,N] matrix per layer)
    Nd = 768
    out = {"xyz": xyz[:, :Nd].contiguous().numpy(), "colors": colors[:, :Nd].contiguous().numpy(),
           "labels": labels[:, :Nd].contiguous().numpy(), "init_seed": np.int64(INIT_SEED),
           "fwd_seed": np.int64(FWD_SEED)}
    for k in (20, 8):
        ref = pair(lambda: RefDGCNN(5, k=k), lambda: DGCNN(5, k=k))
        r = _run_seg(ref, (xyz[:, :Nd].contiguous(), colors[:, :Nd].contiguous()),
                     labels[:, :Nd].contiguous(), 2)
        out.update({f"k{k}_{a}": b for a, b in r.items()})
        print("dgcnn k", k, r["loss"])
    np.savez_compressed(os.path.join(HERE, "model_dgcnn.npz"), **out)
