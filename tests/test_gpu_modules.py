"""GPU: the drop-in modules and the assembled networks against the reference's golden outputs.

Same seeded parameters (the modules create theirs in the reference's order), same inputs, same CPU
generator seed in front of each forward (farthest_point_sample draws its start index there).
Bar from BASELINE.json: segmentation logits within 1e-4 relative (max |diff| / max |reference|).
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.helpers import load_golden

pytestmark = pytest.mark.gpu

REL = 1e-4


def rel_err(got, ref):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12))


def dev(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.requires_grad_(True) if grad else t


def build(cls, seed, *a, **kw):
    torch.manual_seed(int(seed))
    return cls(*a, **kw).cuda()


def grad_norms(m):
    return np.array([0.0 if p.grad is None else float(p.grad.norm()) for _, p in m.named_parameters()])


def assert_grad_norms(got, ref, rtol):
    """Per-parameter gradient L2 norms.  Biases in front of a train-mode BatchNorm have an exactly
    zero gradient in real arithmetic; what either side reports there is rounding noise, hence the
    absolute floor relative to the largest norm."""
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=1e-4 * float(np.max(ref)))


def dropout_eval(m):
    for s in m.modules():
        if isinstance(s, nn.Dropout):
            s.eval()


@pytest.fixture(scope="module")
def mpu():
    assert torch.cuda.is_available()
    from pointcloud_bridge_amd.models import pointnet2_utils
    return pointnet2_utils


@pytest.mark.parametrize("tag,cls,args", [
    ("sa", "SetAbstraction", (128, 0.3, 16, 8, [16, 16, 32])),
    ("msg", "MultiScaleSetAbstraction", (128, [0.2, 0.4], [8, 16], 8, [16, 16, 32])),
])
def test_set_abstraction_modules(mpu, tag, cls, args):
    g = load_golden("modules")
    mod = build(getattr(mpu, cls), g["init_seed"], *args)
    xyz = dev(g["xyz"])
    for mode in ("eval", "train"):
        mod.train(mode == "train")
        f = dev(g["feats"], grad=True)
        torch.manual_seed(int(g["fwd_seed"]))
        new_xyz, out = mod(xyz, f)
        assert np.array_equal(new_xyz.cpu().numpy(), g[f"{tag}_{mode}_new_xyz"])  # same samples, bitwise
        assert tuple(out.shape) == g[f"{tag}_{mode}_out"].shape
        assert rel_err(out, g[f"{tag}_{mode}_out"]) < REL
    mod.zero_grad()
    (out * torch.linspace(-1, 1, out.numel(), device="cuda").view(out.shape)).sum().backward()
    assert rel_err(f.grad, g[f"{tag}_grad_feats"]) < 1e-3
    assert_grad_norms(grad_norms(mod), g[f"{tag}_grad_norms"], 2e-3)


def test_set_abstraction_without_features(mpu):
    g = load_golden("modules")
    mod = build(mpu.SetAbstraction, g["init_seed"], 64, 0.4, 8, 3, [8, 16]).eval()
    torch.manual_seed(int(g["fwd_seed"]))
    _, out = mod(dev(g["xyz"]), None)
    assert rel_err(out, g["sa0_eval_out"]) < REL


@pytest.mark.parametrize("tag,cls,args,use_p1", [
    ("fp", "FeaturePropagation", (18, [16, 8]), True),
    ("fp_nop1", "FeaturePropagation", (12, [16]), False),
    ("efp", "EnhancedFeaturePropagation", (18, [16, 8]), True),
    ("efp_skip", "EnhancedFeaturePropagation", (18, [16, 18]), True),
])
def test_feature_propagation_modules(mpu, tag, cls, args, use_p1):
    g = load_golden("modules")
    mod = build(getattr(mpu, cls), g["init_seed"], *args)
    xyz, xyz2 = dev(g["xyz"]), dev(g["fp_xyz2"])
    for mode in ("eval", "train"):
        mod.train(mode == "train")
        p1 = dev(g["fp_p1"], grad=True) if use_p1 else None
        p2 = dev(g["fp_p2"], grad=True)
        out = mod(xyz, xyz2, p1, p2)
        assert tuple(out.shape) == g[f"{tag}_{mode}_out"].shape
        assert rel_err(out, g[f"{tag}_{mode}_out"]) < REL
    mod.zero_grad()
    (out * torch.linspace(-1, 1, out.numel(), device="cuda").view(out.shape)).sum().backward()
    assert rel_err(p2.grad, g[f"{tag}_grad_p2"]) < 1e-3
    if use_p1:
        assert rel_err(p1.grad, g[f"{tag}_grad_p1"]) < 1e-3
    assert_grad_norms(grad_norms(mod), g[f"{tag}_grad_norms"], 2e-3)


def test_feature_propagation_single_centroid_raises_like_reference(mpu):
    g = load_golden("modules")
    assert int(g["fp_s1_raises"]) == 1
    mod = build(mpu.FeaturePropagation, g["init_seed"], 18, [8]).eval()
    with pytest.raises(RuntimeError):
        mod(dev(g["xyz"]), dev(g["fp_xyz2"][:, :1]), dev(g["fp_p1"]), dev(g["fp_p2"][:, :, :1]))


def test_state_dict_keys_match_reference_layout(mpu):
    sa = mpu.SetAbstraction(8, 0.1, 4, 6, [8, 8])
    assert list(sa.state_dict())[:7] == [
        "mlp_convs.0.weight", "mlp_convs.0.bias", "mlp_convs.1.weight", "mlp_convs.1.bias",
        "mlp_bns.0.weight", "mlp_bns.0.bias", "mlp_bns.0.running_mean"]
    msg = mpu.MultiScaleSetAbstraction(8, [0.1, 0.2], [4, 8], 6, [8])
    assert "conv_blocks.1.0.weight" in msg.state_dict() and "bn_blocks.0.0.num_batches_tracked" in msg.state_dict()
    efp = mpu.EnhancedFeaturePropagation(16, [8])
    assert {"attention.0.weight", "attention.3.bias", "boundary_aware.3.weight"} <= set(efp.state_dict())
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    assert len(DGCNN(5).state_dict()) == 76  # every shared BatchNorm appears twice, as in the reference


def run_seg(model, xyz, colors, labels, fwd_seed, channel_dim):
    model.eval()
    torch.manual_seed(fwd_seed)
    with torch.no_grad():
        le = model(xyz, colors)
    model.train()
    dropout_eval(model)
    torch.manual_seed(fwd_seed)
    lt = model(xyz, colors)
    lg = lt if channel_dim == 1 else lt.reshape(-1, lt.shape[-1])
    lb = labels if channel_dim == 1 else labels.reshape(-1)
    loss = F.cross_entropy(lg, lb)
    model.zero_grad()
    loss.backward()
    return le, lt, float(loss.detach())


@pytest.mark.parametrize("name,kw", [
    ("model_pn2_ssg", dict(cls="PointNet2", rgb_skip=False)),
    ("model_pn2_ssg_skip", dict(cls="PointNet2", rgb_skip=True)),
    ("model_pn2_msg", dict(cls="PointNet2MSG")),
])
def test_pointnet2_networks_logit_parity(name, kw):
    from pointcloud_bridge_amd.models import containers
    g = load_golden(name)
    kw = dict(kw)
    cls = getattr(containers, kw.pop("cls"))
    model = build(cls, g["init_seed"], 5, **kw)
    le, lt, loss = run_seg(model, dev(g["xyz"]), dev(g["colors"]), dev(g["labels"]), int(g["fwd_seed"]), 1)
    assert tuple(le.shape) == g["logits_eval"].shape  # [B, classes, N]
    assert rel_err(le, g["logits_eval"]) < REL
    assert rel_err(lt, g["logits_train"]) < REL
    assert abs(loss - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert_grad_norms(grad_norms(model), g["grad_norms"], 5e-3)


@pytest.mark.parametrize("k", [20, 8])
def test_dgcnn_logit_parity(k):
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    g = load_golden("model_dgcnn")
    model = build(DGCNN, g["init_seed"], 5, k=k)
    le, lt, loss = run_seg(model, dev(g["xyz"]), dev(g["colors"]), dev(g["labels"]), int(g["fwd_seed"]), 2)
    assert tuple(le.shape) == g[f"k{k}_logits_eval"].shape  # [B, N, classes]
    # The graph is rebuilt in 64-d feature space three times.  With the MLPs on the exact fp32 matrix
    # cores every logit of this fixture is within 1e-4 (measured 1.4e-6 eval, 6e-6 train; round 1,
    # with library GEMMs in front of the kNN, had neighbour swaps at last-bit ties move single points
    # by up to 5e-2).  A swap at an exact tie remains possible on other inputs: tests/test_gpu_ops.py
    # pins the kNN itself, tie-tolerantly.
    assert rel_err(le, g[f"k{k}_logits_eval"]) < REL
    assert rel_err(lt, g[f"k{k}_logits_train"]) < REL
    assert abs(loss - float(g[f"k{k}_loss"])) < 1e-5 * abs(float(g[f"k{k}_loss"]))
    assert_grad_norms(grad_norms(model), g[f"k{k}_grad_norms"], 5e-3)


def test_sampling_prefetch_gives_identical_results(mpu):
    """The prefetched FPS pyramid (side stream, drawn earlier from the CPU generator) must reproduce
    the inline result bit for bit, including the RNG stream."""
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    g = load_golden("model_pn2_msg")
    model = build(PointNet2MSG, g["init_seed"], 5).eval()
    xyz, colors = dev(g["xyz"]), dev(g["colors"])
    with torch.no_grad():
        torch.manual_seed(int(g["fwd_seed"]))
        inline = model(xyz, colors)
        after_inline = torch.rand(1)
        torch.manual_seed(int(g["fwd_seed"]))
        model.prefetch(xyz)
        pre = model(xyz, colors)
        after_pre = torch.rand(1)
    assert torch.equal(inline, pre)
    assert torch.equal(after_inline, after_pre)          # same number of CPU draws consumed
    assert rel_err(pre, g["logits_eval"]) < REL
    assert len(model.sampling.prefetched) == 0           # every level was picked up (the model's own SamplingState)
    assert len(mpu._default_state.prefetched) == 0       # ... and nothing leaked into the process default


def test_pipelined_inference_gives_identical_results(mpu):
    """set_next(): the following batch's FPS pyramid starts beside the current pass's decoder.  Two
    consecutive eval passes must produce exactly what two plain passes produce (same CPU-generator
    draws in the same order, same samples)."""
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    g = load_golden("model_pn2_msg")
    xyz, colors = dev(g["xyz"]), dev(g["colors"])
    xyz2 = (xyz * 0.9).contiguous()
    model = build(PointNet2MSG, g["init_seed"], 5).eval()
    for mode in ("fp32", "bf16"):
        with rowmlp.precision(mode):
            xyz3 = (xyz * 0.8).contiguous()
            with torch.no_grad():
                torch.manual_seed(11)
                plain = [model(c, colors) for c in (xyz, xyz2, xyz3)]
                torch.manual_seed(11)
                piped = []
                for cur, nxt in ((xyz, xyz2), (xyz2, xyz3), (xyz3, None)):
                    if nxt is not None:
                        model.set_next(nxt)  # first pass: started behind its encoder; second: at its top
                    piped.append(model(cur, colors))
            assert all(torch.equal(a, b) for a, b in zip(plain, piped)), mode


def test_flat_adam_equals_torch_fused_adam():
    """parallel.FlatAdam (one flat buffer, one fused launch) == torch.optim.Adam(fused=True), bit for bit."""
    from pointcloud_bridge_amd.parallel import FlatAdam
    dev = torch.device("cuda")
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(13, 32), torch.nn.BatchNorm1d(32), torch.nn.Linear(32, 5)).to(dev)
    mine = copy.deepcopy(ref)
    o_ref = torch.optim.Adam(ref.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4, fused=True)
    o_mine = FlatAdam(mine.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4)
    x = torch.randn(64, 13, device=dev)
    for _ in range(5):
        for m, o in ((ref, o_ref), (mine, o_mine)):
            o.zero_grad(set_to_none=True)
            m(x).square().mean().backward()
            o.step()
    for a, b in zip(ref.parameters(), mine.parameters()):
        assert torch.equal(a, b)


def test_dgcnn_graph_prefetch_gives_identical_results():
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    g = load_golden("model_dgcnn")
    xyz, colors = dev(g["xyz"]), dev(g["colors"])
    model = build(DGCNN, g["init_seed"], 5, k=20).eval()
    with torch.no_grad():
        plain = model(xyz, colors)
        model.prefetch(xyz)
        ahead = model(xyz, colors)
        again = model(xyz, colors)   # the parked graph is used once
    assert torch.equal(plain, ahead) and torch.equal(plain, again)
