"""Generate golden vectors by running the REFERENCE implementation on CPU.

Run in the build container only (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [ops|modules|models|all]

Writes small .npz fixtures next to this file.  A fixture holds inputs and the
reference's outputs only -- no reference source text.  The reference modules are
imported from /root/reference/Highway_bridge (models/pointnet2_utils.py,
models/DGCNN.py, models/model.py, models/pointnet2.py) with torch 2.10 fp32 on CPU.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Highway_bridge"


def _ref():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models import pointnet2_utils as pu  # noqa
    from models.DGCNN import DGCNN  # noqa
    return pu, DGCNN


def unit_ball_cloud(gen, B, N):
    """Uniform in the unit ball, then centred / max-norm scaled like utils/simpdataset.py:47-62."""
    v = torch.randn(B, N, 3, generator=gen)
    v = v / v.norm(dim=-1, keepdim=True)
    r = torch.rand(B, N, 1, generator=gen) ** (1.0 / 3.0)
    p = v * r
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)
    return p.contiguous()


def grid_cloud(gen, B, N):
    """Coordinates on a 2^-10 grid in [-1,1): every fp32 distance is exact."""
    return (torch.randint(-1024, 1024, (B, N, 3), generator=gen).float() / 1024.0).contiguous()


def dup_cloud(gen, B, N, uniq):
    """Short cloud padded by repeating points (utils/simpdataset.py:146-148)."""
    base = unit_ball_cloud(gen, B, uniq)
    extra = torch.randint(0, uniq, (B, N - uniq), generator=gen)
    rep = torch.gather(base, 1, extra.unsqueeze(-1).expand(-1, -1, 3))
    return torch.cat([base, rep], dim=1).contiguous()


def ops_case(pu, DGCNN, xyz, S, balls, knn_k, seed):
    """Run every index op of the reference on one cloud."""
    B, N, _ = xyz.shape
    out = {"xyz": xyz.numpy()}
    torch.manual_seed(seed)
    state = torch.get_rng_state()
    start = torch.randint(0, N, (B,), dtype=torch.long)  # what pointnet2_utils.py:69 will draw
    torch.set_rng_state(state)
    fps = pu.farthest_point_sample(xyz, S)
    assert torch.equal(fps[:, 0], start)
    out["fps_start"] = start.numpy()
    out["fps_idx"] = fps.numpy()
    new_xyz = pu.index_points(xyz, fps)
    out["new_xyz"] = new_xyz.numpy()
    for t, (r, ns) in enumerate(balls):
        out[f"ball{t}_r"] = np.float64(r)
        out[f"ball{t}_ns"] = np.int64(ns)
        out[f"ball{t}_idx"] = pu.query_ball_point(r, ns, xyz, new_xyz).numpy()
    d = pu.square_distance(xyz, new_xyz)
    ds, di = d.sort(dim=-1)
    out["nn_d"] = ds[:, :, :4].contiguous().numpy()
    out["nn_idx"] = di[:, :, :4].contiguous().numpy()
    net = DGCNN.__new__(DGCNN)  # knn/get_graph_feature use no parameters
    x = xyz.transpose(1, 2).contiguous()
    idx = DGCNN.knn(net, x, knn_k)
    out["knn_k"] = np.int64(knn_k)
    out["knn_idx"] = idx.numpy()
    return out


def make_ops():
    pu, DGCNN = _ref()
    g = torch.Generator().manual_seed(0)
    cases = {
        "ops_grid": ops_case(pu, DGCNN, grid_cloud(g, 2, 1024), 128, [(0.2, 16), (0.4, 32)], 20, 1),
        "ops_cont": ops_case(pu, DGCNN, unit_ball_cloud(g, 2, 2048), 256, [(0.1, 16), (0.2, 32)], 20, 2),
        "ops_dup": ops_case(pu, DGCNN, dup_cloud(g, 2, 512, 300), 64, [(0.2, 16), (0.4, 32)], 8, 3),
        # nearly as few points as nsample (the reference raises IndexError once nsample > N) and a radius that only the centroid satisfies
        "ops_tiny": ops_case(pu, DGCNN, unit_ball_cloud(g, 3, 24), 8, [(1e-3, 4), (5.0, 16)], 5, 4),
    }
    for name, c in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **c)
        print(name, {k: getattr(v, "shape", v) for k, v in c.items()})

    # square_distance itself (small, fully materialised) + centroids that see no point at all
    xyz = unit_ball_cloud(g, 2, 256)
    far = unit_ball_cloud(g, 2, 16) + 10.0
    sd = pu.square_distance(far[:, :8] - 10.0, xyz)
    empty = pu.query_ball_point(0.3, 8, xyz, far)
    # kNN in feature space (D = 64) and the EdgeConv gather
    feat = torch.randn(2, 64, 384, generator=g)
    net = DGCNN.__new__(DGCNN)
    kidx = DGCNN.knn(net, feat, 20)
    graph = DGCNN.get_graph_feature(net, feat[:, :8, :64].contiguous(), k=5)
    gidx = DGCNN.knn(net, feat[:, :8, :64].contiguous(), 5)
    # 3-NN interpolation (k = 3) and 4-NN (EnhancedFeaturePropagation) on features
    xyz1 = unit_ball_cloud(g, 2, 512)
    xyz2 = xyz1[:, :96].contiguous()
    p2 = torch.randn(2, 96, 32, generator=g)
    res = {}
    for k in (3, 4):
        d = pu.square_distance(xyz1, xyz2)
        ds, di = d.sort(dim=-1)
        ds, di = ds[:, :, :k], di[:, :, :k]
        rec = 1.0 / (ds + 1e-8)
        w = rec / torch.sum(rec, dim=2, keepdim=True)
        interp = torch.sum(pu.index_points(p2, di) * w.view(2, 512, k, 1), dim=2)
        res[f"interp{k}_d"] = ds.contiguous().numpy()
        res[f"interp{k}_idx"] = di.contiguous().numpy()
        res[f"interp{k}_w"] = w.numpy()
        res[f"interp{k}_out"] = interp.numpy()
    np.savez_compressed(
        os.path.join(HERE, "ops_misc.npz"),
        sd_src=(far[:, :8] - 10.0).numpy(), sd_dst=xyz.numpy(), sd_out=sd.numpy(),
        empty_xyz=xyz.numpy(), empty_new_xyz=far.numpy(), empty_idx=empty.numpy(),
        knn64_x=feat.numpy(), knn64_idx=kidx.numpy(),
        graph_x=feat[:, :8, :64].contiguous().numpy(), graph_idx=gidx.numpy(), graph_out=graph.numpy(),
        interp_xyz1=xyz1.numpy(), interp_xyz2=xyz2.numpy(), interp_p2=p2.numpy(), **res)
    print("ops_misc written")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.set_num_threads(8)
    if what in ("ops", "all"):
        make_ops()
    if what in ("modules", "all"):
        from make_golden_modules import make_modules
        make_modules()
    if what in ("models", "all"):
        from make_golden_modules import make_models
        make_models()


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    sys.path.insert(0, REPO)
    main()
