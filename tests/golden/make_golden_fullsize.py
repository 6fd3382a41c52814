"""Full-size golden vectors from the REFERENCE (CPU, fp32): the networks of BASELINE.json configs[1] and configs[2]
at their real cloud sizes, one scene each (the reference's ball query materialises [B,S,N] int64 + a full sort and its
kNN a [B,N,N] matrix per layer: B = 1 keeps the generator at a few GB and a few seconds per pass).

  model_pn2_msg_full.npz   PointNet++ MSG trunk (models/model.py:58-147 without the bridge encoders, as
                           make_golden_modules._RefMsgTrunk), B=1 x N=16384: eval + train logits, loss, gradient norms,
                           AND the index tensors the reference computed on the way -- farthest_point_sample of the
                           three levels (models/pointnet2_utils.py:63-80) and query_ball_point of every (level, radius)
                           (:97-112), recorded by wrapping the reference's own functions while its forward pass runs.
  model_dgcnn_full.npz     DGCNN(5, k=20) (models/DGCNN.py:111-172), B=1 x N=8192: the same, with the kNN lists of the
                           four EdgeConv blocks (:49-70).

    python tests/golden/make_golden_fullsize.py        (here, in the build container; needs /root/reference)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden_modules as mgm  # noqa: E402  (puts the reference and the repo on sys.path)
from make_golden import unit_ball_cloud  # noqa: E402


def _record(module, name, log):
    """Wrap module.name so that every result is appended to log (the reference calls these through its module globals)."""
    orig = getattr(module, name)

    def wrapped(*a, **kw):
        out = orig(*a, **kw)
        log.append(out.detach().clone())
        return out

    setattr(module, name, wrapped)
    return lambda: setattr(module, name, orig)


def small_int(t, bound):
    a = t.numpy()
    assert a.min() >= 0 and a.max() < bound
    return a.astype(np.uint16 if bound <= 65536 else np.int32)


def pn2_msg():
    from models import pointnet2_utils as rpu
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    g = torch.Generator().manual_seed(2024)
    B, N = 1, 16384
    xyz = unit_ball_cloud(g, B, N)
    colors = torch.rand(B, N, 3, generator=g)
    labels = torch.randint(0, 5, (B, N), generator=g)
    torch.manual_seed(mgm.INIT_SEED)
    ref = mgm._RefMsgTrunk(5)
    torch.manual_seed(mgm.INIT_SEED)
    mgm._same_params(ref, PointNet2MSG(5))
    fps, ball = [], []
    undo = [_record(rpu, "farthest_point_sample", fps), _record(rpu, "query_ball_point", ball)]
    try:
        out = mgm._run_seg(ref, (xyz, colors), labels, 1)
    finally:
        for u in undo:
            u()
    # _run_seg makes two passes (eval, train) under the same CPU-generator seed: identical index tensors
    assert len(fps) == 6 and len(ball) == 12
    for a, b in zip(fps[:3] + ball[:6], fps[3:] + ball[6:]):
        assert torch.equal(a, b)
    sizes = [N, 1024, 512]
    for lvl in range(3):
        out[f"fps{lvl + 1}"] = small_int(fps[lvl], sizes[lvl])
        for r in range(2):
            out[f"ball{lvl + 1}_{r}"] = small_int(ball[2 * lvl + r], sizes[lvl])
    out.update({"xyz": xyz.numpy(), "colors": colors.numpy(), "labels": labels.numpy().astype(np.uint8),
                "init_seed": np.int64(mgm.INIT_SEED), "fwd_seed": np.int64(mgm.FWD_SEED)})
    np.savez_compressed(os.path.join(HERE, "model_pn2_msg_full.npz"), **out)
    print("model_pn2_msg_full", out["loss"], out["logits_train"].shape, {k: v.shape for k, v in out.items() if k.startswith(("fps", "ball"))})


def dgcnn():
    from models.DGCNN import DGCNN as RefDGCNN
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    g = torch.Generator().manual_seed(2025)
    B, N, k = 1, 8192, 20
    xyz = unit_ball_cloud(g, B, N)
    colors = torch.rand(B, N, 3, generator=g)
    labels = torch.randint(0, 5, (B, N), generator=g)
    torch.manual_seed(mgm.INIT_SEED)
    ref = RefDGCNN(5, k=k)
    torch.manual_seed(mgm.INIT_SEED)
    mgm._same_params(ref, DGCNN(5, k=k))
    knn = []
    orig = ref.knn

    def wrapped(x, kk):
        idx = orig(x, kk)
        knn.append(idx.detach().clone())
        return idx

    ref.knn = wrapped
    out = mgm._run_seg(ref, (xyz, colors), labels, 2)
    assert len(knn) == 8                              # four graphs per pass, eval pass first
    for i in range(4):
        out[f"knn_eval{i + 1}"] = small_int(knn[i], N)
        out[f"knn_train{i + 1}"] = small_int(knn[4 + i], N)
    out.update({"xyz": xyz.numpy(), "colors": colors.numpy(), "labels": labels.numpy().astype(np.uint8),
                "init_seed": np.int64(mgm.INIT_SEED), "fwd_seed": np.int64(mgm.FWD_SEED), "k": np.int64(k)})
    np.savez_compressed(os.path.join(HERE, "model_dgcnn_full.npz"), **out)
    print("model_dgcnn_full", out["loss"], out["logits_train"].shape)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["pn2_msg", "dgcnn"]
    if "pn2_msg" in which:
        pn2_msg()
    if "dgcnn" in which:
        dgcnn()
