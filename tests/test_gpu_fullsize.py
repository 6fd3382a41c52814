"""GPU: the networks of BASELINE.json configs[1] / configs[2] at their REAL cloud sizes against reference fixtures
(round 3; tests/golden/make_golden_fullsize.py ran the reference on CPU: one scene of 16384 / 8192 points).

Until this round full size was covered by properties of the index operators only (tests/test_gpu_ops.py); the network
goldens stopped at N = 2048.  Here, for PointNet++ MSG at N = 16384: every index tensor the reference computed on its
way -- three farthest_point_sample calls (models/pointnet2_utils.py:63-80), six query_ball_point calls (:97-112) -- must
be reproduced bit for bit, fp32 rows give the logits within 1e-4 (north_star), bf16 rows within measured bars.  For
DGCNN k=20 at N = 8192: the four kNN graphs (models/DGCNN.py:49-70) and the logits.
"""
import numpy as np
import pytest
import torch

from tests.helpers import load_golden
from tests.test_gpu_modules import assert_grad_norms, build, dev, grad_norms, rel_err, run_seg

pytestmark = pytest.mark.gpu


def _labels(g):
    return torch.from_numpy(g["labels"].astype(np.int64)).cuda()


def test_pn2_msg_indices_at_n16384_are_the_reference_s():
    """FPS of the three levels and the ball queries of every (level, radius), chained exactly as
    MultiScaleSetAbstraction.forward chains them (:326-347), under the same CPU-generator seed."""
    from pointcloud_bridge_amd.models import pointnet2_utils as mpu
    g = load_golden("model_pn2_msg_full")
    xyz = dev(g["xyz"])
    levels = [(1024, [0.1, 0.2], [16, 32]), (512, [0.2, 0.4], [16, 32]), (128, [0.4, 0.8], [16, 32])]
    torch.manual_seed(int(g["fwd_seed"]))
    cur = xyz
    for lvl, (npoint, radii, nsamples) in enumerate(levels, 1):
        fps = mpu.farthest_point_sample(cur, npoint)
        assert fps.dtype == torch.int64
        assert np.array_equal(fps.cpu().numpy(), g[f"fps{lvl}"].astype(np.int64)), f"fps level {lvl}"
        new_xyz = mpu.index_points(cur, fps)
        for r, (radius, ns) in enumerate(zip(radii, nsamples)):
            ball = mpu.query_ball_point(radius, ns, cur, new_xyz)
            assert np.array_equal(ball.cpu().numpy(), g[f"ball{lvl}_{r}"].astype(np.int64)), f"ball level {lvl} radius {radius}"
        cur = new_xyz


def test_pn2_msg_logits_at_n16384_fp32_rows():
    """north_star's bar at the benchmark's cloud size: eval and train logits within 1e-4 relative, loss, gradient norms."""
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    g = load_golden("model_pn2_msg_full")
    model = build(PointNet2MSG, g["init_seed"], 5)
    le, lt, loss = run_seg(model, dev(g["xyz"]), dev(g["colors"]), _labels(g), int(g["fwd_seed"]), 1)
    print("pn2_msg N=16384 fp32 rows: eval", rel_err(le, g["logits_eval"]), "train", rel_err(lt, g["logits_train"]),
          "loss", abs(loss - float(g["loss"])) / float(g["loss"]))
    assert rel_err(le, g["logits_eval"]) < 1e-4
    assert rel_err(lt, g["logits_train"]) < 1e-4
    assert abs(loss - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert_grad_norms(grad_norms(model), g["grad_norms"], 5e-3)


# Measured on MI355X (round 3, centred bf16 rows), max |d| / max |ref| and mean |d| / mean |ref|:
#   pn2_msg N=16384: eval 5.5e-3 / 1.9e-3, train 1.8e-1 / 1.5e-1, loss 7e-5, gradient norms median 2.2e-2, max 0.18
#   (fp32 rows, the test above: eval 7.7e-7, train 3.2e-5, loss 7e-8)
# (the train-mode distance is the network's own noise amplification: tests/test_gpu_round2.py, tools/bf16_mixed.py)
_BF16_FULL = dict(eval_max=1.2e-2, eval_mean=5e-3, train_max=0.36, train_mean=0.3, loss=3e-3, gn_median=4.5e-2, gn_max=0.4)


def test_pn2_msg_logits_at_n16384_bf16_rows():
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    g = load_golden("model_pn2_msg_full")
    model = build(PointNet2MSG, g["init_seed"], 5)
    with rowmlp.precision("bf16"):
        le, lt, loss = run_seg(model, dev(g["xyz"]), dev(g["colors"]), _labels(g), int(g["fwd_seed"]), 1)
    e = {}
    for tag, got, ref in (("eval", le, g["logits_eval"]), ("train", lt, g["logits_train"])):
        d = np.abs(got.float().detach().cpu().numpy() - ref)
        e[f"{tag}_max"], e[f"{tag}_mean"] = float(d.max() / np.abs(ref).max()), float(d.mean() / np.abs(ref).mean())
    e["loss"] = abs(loss - float(g["loss"])) / abs(float(g["loss"]))
    gn, ref = grad_norms(model), g["grad_norms"]
    big = ref > 1e-3 * ref.max()
    r = np.abs(gn[big] - ref[big]) / ref[big]
    e["gn_median"], e["gn_max"] = float(np.median(r)), float(r.max())
    print("pn2_msg N=16384 bf16 rows:", {k: f"{v:.3e}" for k, v in e.items()})
    for key, bar in _BF16_FULL.items():
        assert e[key] < bar, (key, e[key], bar)


def _same_sets(a, b):
    """Fraction of rows of two neighbour tables [B,N,k] that hold the same index SET (ties permute freely)."""
    return float((np.sort(a, axis=-1) == np.sort(b, axis=-1)).all(axis=-1).mean())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dgcnn_at_n8192(precision, monkeypatch):
    """DGCNN k=20, one scene of 8192 points.  The coordinate graph is the reference's list exactly; the three
    feature-space graphs are built from features that agree with the reference's to ~1e-6 (fp32 rows), so a neighbour
    at a near-tie may swap: the fraction of identical rows is measured and bounded.  Logits: fp32 rows within 1e-4 on
    the bulk (a swapped neighbour moves single points), bf16 rows: measured bars."""
    from pointcloud_bridge_amd import ops, rowmlp
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    g = load_golden("model_dgcnn_full")
    k = int(g["k"])
    model = build(DGCNN, g["init_seed"], 5, k=k)
    seen = []
    real = ops.knn

    def recording(x, kk, *a, **kw):
        idx = real(x, kk, *a, **kw)
        seen.append(idx.detach().cpu().numpy())
        return idx

    monkeypatch.setattr(ops, "knn", recording)
    with rowmlp.precision(precision):
        le, lt, loss = run_seg(model, dev(g["xyz"]), dev(g["colors"]), _labels(g), int(g["fwd_seed"]), 2)
    assert len(seen) == 8
    same = {}
    for i in range(4):
        same[f"eval{i + 1}"] = _same_sets(seen[i], g[f"knn_eval{i + 1}"].astype(np.int64))
        same[f"train{i + 1}"] = _same_sets(seen[4 + i], g[f"knn_train{i + 1}"].astype(np.int64))
    out = {}
    for tag, got, ref in (("eval", le, g["logits_eval"]), ("train", lt, g["logits_train"])):
        d = np.abs(got.float().detach().cpu().numpy() - ref)
        out[f"{tag}_max"], out[f"{tag}_mean"] = float(d.max() / np.abs(ref).max()), float(d.mean() / np.abs(ref).mean())
        out[f"{tag}_within_1e-4"] = float(np.mean(d / np.abs(ref).max() < 1e-4))
    out["loss"] = abs(loss - float(g["loss"])) / abs(float(g["loss"]))
    print(f"dgcnn N=8192 {precision} rows: graphs identical rows", {a: f"{b:.5f}" for a, b in same.items()},
          {a: f"{b:.3e}" for a, b in out.items()})
    assert same["eval1"] == 1.0 and same["train1"] == 1.0        # coordinates only: the reference's lists
    if precision == "fp32":
        # measured (MI355X, round 3): identical rows 1.0 / 1.0 / 0.99988 / 0.99963 (eval), 1.0 / 0.99988 / 0.99951 /
        # 0.99670 (train); eval logits 99.99 % within 1e-4 (mean 1.3e-6, max 2.4e-4), train 98.1 % (mean 6.6e-4, max
        # 5.7e-2), loss 2.6e-5.  With 8192 candidates per query, neighbours at relative distance gaps of 1e-6 are common;
        # features that differ from the reference's in the seventh digit (another fp32 summation order) swap them, the
        # point's max-pooled feature moves, and in train mode every swap also nudges the batch statistics.
        assert min(same.values()) > 0.995
        assert out["eval_within_1e-4"] > 0.999 and out["train_within_1e-4"] > 0.97
        assert out["eval_mean"] < 1e-5 and out["train_mean"] < 2e-3 and out["loss"] < 1e-4
    else:
        # measured (MI355X, round 3): feature graphs 60-75 % identical rows (bf16 features move near neighbours),
        # eval 1.1e-2 / 2.6e-3, train max 0.5 / mean 0.3, loss 2e-3 -- a dynamic-graph network is another function
        # under another rounding; see tests/test_gpu_bf16.py::test_bf16_networks_track_fp32_networks
        assert out["eval_mean"] < 1e-2 and out["train_mean"] < 0.6 and out["loss"] < 2e-2
