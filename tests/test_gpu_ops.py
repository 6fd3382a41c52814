"""GPU: every HIP operator (through the C ABI) against the oracle and the reference's golden vectors.

Bars: indices bit-identical (FPS, ball query, 3-NN; kNN up to permutations of exactly equal
distances); fp32 distances bit-identical; interpolated features within 1e-6 relative.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import load_golden, knn_tie_tolerant_mismatch

pytestmark = pytest.mark.gpu

CASES = ["ops_grid", "ops_cont", "ops_dup", "ops_tiny"]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from pointcloud_bridge_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def unit_ball(gen, B, N):
    v = torch.randn(B, N, 3, generator=gen)
    v = v / v.norm(dim=-1, keepdim=True)
    p = v * torch.rand(B, N, 1, generator=gen) ** (1.0 / 3.0)
    p = p - p.mean(dim=1, keepdim=True)
    return (p / p.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)).contiguous()


# ------------------------------------------------------------------ golden vectors (reference)
@pytest.mark.parametrize("name", CASES)
def test_fps_golden(ops, name):
    g = load_golden(name)
    S = g["fps_idx"].shape[1]
    got = ops.furthest_point_sample(dev(g["xyz"]), S, dev(g["fps_start"]))
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), g["fps_idx"])
    new_xyz = ops.gather_rows(dev(g["xyz"]), got)
    assert np.array_equal(new_xyz.cpu().numpy(), g["new_xyz"])


@pytest.mark.parametrize("name", CASES)
def test_ball_query_golden(ops, name):
    g = load_golden(name)
    xyz, new_xyz = dev(g["xyz"]), dev(g["new_xyz"])
    r = [float(g["ball0_r"]), float(g["ball1_r"])]
    ns = [int(g["ball0_ns"]), int(g["ball1_ns"])]
    for t in (0, 1):
        got = ops.ball_query(r[t], ns[t], xyz, new_xyz)
        assert np.array_equal(got.cpu().numpy(), g[f"ball{t}_idx"])
    a, b = ops.ball_query2(r, ns, xyz, new_xyz)
    assert np.array_equal(a.cpu().numpy(), g["ball0_idx"])
    assert np.array_equal(b.cpu().numpy(), g["ball1_idx"])


def test_ball_query_empty_ball_and_errors(ops):
    g = load_golden("ops_misc")
    got = ops.ball_query(0.3, 8, dev(g["empty_xyz"]), dev(g["empty_new_xyz"]))
    assert np.array_equal(got.cpu().numpy(), g["empty_idx"])
    with pytest.raises(IndexError):  # the reference raises IndexError once nsample > N
        ops.ball_query(0.3, 300, dev(g["empty_xyz"]), dev(g["empty_new_xyz"]))
    with pytest.raises(RuntimeError):
        ops.ball_query(0.3, 8, torch.from_numpy(g["empty_xyz"]), torch.from_numpy(g["empty_new_xyz"]))


def test_square_distance_golden_bitwise(ops):
    g = load_golden("ops_misc")
    got = ops.square_distance(dev(g["sd_src"]), dev(g["sd_dst"])).cpu().numpy()
    assert np.array_equal(got.view(np.int32), g["sd_out"].view(np.int32))


@pytest.mark.parametrize("name", CASES)
def test_three_nn_golden(ops, name):
    g = load_golden(name)
    for k in (3, 4):
        d, i = ops.three_nn(dev(g["xyz"]), dev(g["new_xyz"]), k)
        assert np.array_equal(d.cpu().numpy().view(np.int32), g["nn_d"][:, :, :k].view(np.int32))
        assert np.array_equal(i.cpu().numpy(), g["nn_idx"][:, :, :k])  # stable tie order included


@pytest.mark.parametrize("k", [3, 4])
def test_interpolate_golden(ops, k):
    g = load_golden("ops_misc")
    d, i = ops.three_nn(dev(g["interp_xyz1"]), dev(g["interp_xyz2"]), k)
    assert np.array_equal(i.cpu().numpy(), g[f"interp{k}_idx"])
    out = ops.three_interpolate(dev(g["interp_p2"]), d, i)
    np.testing.assert_allclose(out.cpu().numpy(), g[f"interp{k}_out"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_knn_xyz_golden(ops, name):
    g = load_golden(name)
    k = int(g["knn_k"])
    got = ops.knn(dev(g["xyz"]), k).cpu().numpy()
    # bit-identical to the oracle (same arithmetic, same tie rule) ...
    assert np.array_equal(got, orc.knn(g["xyz"], k))
    # ... and equal to the reference up to permutations of (near-)equal distances
    x = g["xyz"]

    def dist_of(b, i, js):
        return orc.square_distance(x[b:b + 1, i:i + 1], x[b:b + 1, js])[0, 0]

    assert knn_tie_tolerant_mismatch(got, g["knn_idx"], dist_of, atol=1e-6) == 0


def test_knn_d64_golden(ops):
    g = load_golden("ops_misc")
    x = np.ascontiguousarray(g["knn64_x"].transpose(0, 2, 1))
    got = ops.knn(dev(x), 20).cpu().numpy()
    assert np.array_equal(got, orc.knn(x, 20))
    # kernel == oracle bit for bit, and the oracle carries the reference's two summation orders
    # (tests/test_oracle_golden.py::test_knn_feature_space_d64): the reference's lists, up to exact ties
    B, N, _ = x.shape
    full_i, full_d = orc.knn(x, N, return_dist=True)
    dmap = np.empty((B, N, N), np.float32)
    for b in range(B):
        for i in range(N):
            dmap[b, i, full_i[b, i]] = full_d[b, i]
    assert knn_tie_tolerant_mismatch(got, g["knn64_idx"], lambda b, i, js: dmap[b, i, js]) == 0
    assert (got != g["knn64_idx"]).mean() < 1e-4


def test_edge_features_golden(ops):
    g = load_golden("ops_misc")
    x = np.ascontiguousarray(g["graph_x"].transpose(0, 2, 1))
    got = ops.edge_features(dev(x), dev(g["graph_idx"])).permute(0, 3, 1, 2).contiguous()
    assert np.array_equal(got.cpu().numpy(), g["graph_out"])


# ------------------------------------------------------------------ oracle, seeded inputs
@pytest.mark.parametrize("B,N,S", [(1, 1, 1), (2, 63, 17), (3, 65, 64), (2, 200, 50), (2, 513, 128),
                                   (2, 1000, 333), (1, 1500, 300), (2, 3000, 512), (1, 5000, 700),
                                   (2, 12000, 1024), (1, 16384, 1024), (1, 20000, 256)])
def test_fps_vs_oracle_every_kernel_variant(ops, B, N, S):
    gen = torch.Generator().manual_seed(N)
    xyz = unit_ball(gen, B, N) if N > 1 else torch.zeros(B, 1, 3)
    S = min(S, N)
    start = torch.randint(0, N, (B,), generator=gen)
    got = ops.furthest_point_sample(xyz.cuda(), S, start.cuda()).cpu().numpy()
    assert np.array_equal(got, orc.farthest_point_sample(xyz.numpy(), S, start.numpy()))


def test_fps_duplicates_pick_first_index(ops):
    gen = torch.Generator().manual_seed(5)
    base = unit_ball(gen, 2, 100)
    xyz = torch.cat([base, base, base[:, :56]], dim=1).contiguous()  # every point repeated
    start = torch.tensor([3, 250])
    got = ops.furthest_point_sample(xyz.cuda(), 128, start.cuda()).cpu().numpy()
    assert np.array_equal(got, orc.farthest_point_sample(xyz.numpy(), 128, start.numpy()))


@pytest.mark.parametrize("kind,N,S", [("dup", 6000, 300), ("dup", 16384, 200), ("flat", 9000, 256),
                                       ("same", 5000, 40), ("grid", 16384, 400), ("line", 8193, 200),
                                       ("ball", 4097, 128), ("ball", 16383, 512), ("clusters", 12345, 300)])
def test_fps_sorted_kernel_tie_and_degenerate_clouds(ops, kind, N, S):
    """The spatially sorted FPS kernel (N > 4096: Morton-sorted cloud, per-wave bounding-box pruning,
    ties broken on ORIGINAL indices) against the oracle on clouds built to stress it: repeated points,
    a plane (zero extent along z), one repeated point, an exact grid full of equal distances, a
    line, sizes that leave ragged last waves, tight clusters (most waves pruned)."""
    gen = torch.Generator().manual_seed(N + S)
    B = 2
    if kind == "dup":
        base = unit_ball(gen, B, N // 3)
        xyz = torch.cat([base, base, base[:, :N - 2 * (N // 3)]], dim=1)
    elif kind == "flat":
        xyz = unit_ball(gen, B, N)
        xyz[..., 2] = 0.25
    elif kind == "same":
        xyz = torch.full((B, N, 3), 0.5)
    elif kind == "grid":
        xyz = torch.randint(-8, 8, (B, N, 3), generator=gen).float() / 8.0
    elif kind == "line":
        xyz = torch.zeros(B, N, 3)
        xyz[..., 0] = torch.rand(B, N, generator=gen)
    elif kind == "clusters":
        centres = unit_ball(gen, B, 8)
        pick = torch.randint(0, 8, (B, N), generator=gen)
        xyz = torch.gather(centres, 1, pick.unsqueeze(-1).expand(-1, -1, 3)) + 0.01 * torch.randn(B, N, 3, generator=gen)
    else:
        xyz = unit_ball(gen, B, N)
    xyz = xyz.contiguous()
    start = torch.randint(0, N, (B,), generator=gen)
    got = ops.furthest_point_sample(xyz.cuda(), S, start.cuda()).cpu().numpy()
    assert np.array_equal(got, orc.farthest_point_sample(xyz.numpy(), S, start.numpy()))


@pytest.mark.parametrize("B,N,S,r,ns", [(2, 1000, 77, 0.15, 16), (1, 4096, 1024, 0.1, 32),
                                         (2, 16384, 130, 0.05, 64), (1, 333, 333, 2.5, 333),
                                         (2, 70, 9, 0.2, 1)])
def test_ball_query_vs_oracle(ops, B, N, S, r, ns):
    gen = torch.Generator().manual_seed(S)
    xyz = unit_ball(gen, B, N)
    new_xyz = xyz[:, torch.randperm(N, generator=gen)[:S]].contiguous()
    got = ops.ball_query(r, ns, xyz.cuda(), new_xyz.cuda()).cpu().numpy()
    assert np.array_equal(got, orc.query_ball_point(r, ns, xyz.numpy(), new_xyz.numpy()))
    a, b = ops.ball_query2([r, 2 * r], [ns, max(1, ns // 2)], xyz.cuda(), new_xyz.cuda())
    assert np.array_equal(a.cpu().numpy(), got)
    assert np.array_equal(b.cpu().numpy(), orc.query_ball_point(2 * r, max(1, ns // 2), xyz.numpy(), new_xyz.numpy()))


@pytest.mark.parametrize("B,N,S,k", [(2, 1000, 77, 3), (1, 16384, 1024, 3), (2, 300, 2500, 4), (1, 5, 4, 4),
                                      (2, 100, 3, 3)])
def test_three_nn_vs_oracle(ops, B, N, S, k):
    gen = torch.Generator().manual_seed(N + S)
    xyz1, xyz2 = unit_ball(gen, B, N), unit_ball(gen, B, S)
    d, i = ops.three_nn(xyz1.cuda(), xyz2.cuda(), k)
    od, oi = orc.three_nn(xyz1.numpy(), xyz2.numpy(), k)
    assert np.array_equal(i.cpu().numpy(), oi)
    assert np.array_equal(d.cpu().numpy().view(np.int32), od.view(np.int32))


@pytest.mark.parametrize("B,N,D,k", [(2, 700, 3, 20), (1, 2048, 3, 8), (2, 600, 64, 20), (1, 300, 64, 32),
                                      (2, 257, 10, 5), (1, 200, 128, 16), (1, 40, 33, 40 - 8)])
def test_knn_vs_oracle(ops, B, N, D, k):
    gen = torch.Generator().manual_seed(N + D)
    x = torch.randn(B, N, D, generator=gen)
    got = ops.knn(x.cuda(), k).cpu().numpy()
    assert np.array_equal(got, orc.knn(x.numpy(), k))


def test_knn_duplicate_points_ties_by_lower_index(ops):
    gen = torch.Generator().manual_seed(9)
    base = torch.randn(1, 150, 3, generator=gen)
    x = torch.cat([base, base], dim=1).contiguous()
    got = ops.knn(x.cuda(), 8).cpu().numpy()
    assert np.array_equal(got, orc.knn(x.numpy(), 8))


# ------------------------------------------------------------------ gathers and their backwards
def test_gather_rows_clamps_like_reference(ops):
    gen = torch.Generator().manual_seed(1)
    pts = torch.randn(2, 50, 7, generator=gen)
    idx = torch.randint(-5, 60, (2, 9, 4), generator=gen)
    got = ops.gather_rows(pts.cuda(), idx.cuda()).cpu().numpy()
    assert got.shape == (2, 9, 4, 7)
    assert np.array_equal(got, orc.index_points(pts.numpy(), idx.numpy()))


def test_group_points_forward_backward(ops):
    gen = torch.Generator().manual_seed(2)
    B, N, S, ns, C = 2, 300, 40, 8, 5
    xyz = unit_ball(gen, B, N).cuda()
    new_xyz = xyz[:, :S].contiguous()
    feat = torch.randn(B, N, C, generator=gen).cuda().requires_grad_(True)
    idx = torch.randint(0, N, (B, S, ns), generator=gen).cuda()
    out = ops.group_points(xyz, new_xyz, feat, idx)
    bi = torch.arange(B, device="cuda").view(B, 1, 1)
    ref = torch.cat([xyz[bi, idx] - new_xyz.view(B, S, 1, 3), feat[bi, idx]], dim=-1)
    assert torch.equal(out, ref)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    g_hip = feat.grad.clone()
    feat.grad = None
    (ref * w).sum().backward()
    torch.testing.assert_close(g_hip, feat.grad, rtol=1e-5, atol=1e-5)
    # no features: only the centred coordinates
    out0 = ops.group_points(xyz, new_xyz, None, idx)
    assert torch.equal(out0, ref[..., :3])


def test_interpolate_backward_matches_autograd(ops):
    gen = torch.Generator().manual_seed(3)
    B, N, S, C = 2, 257, 31, 9
    xyz1, xyz2 = unit_ball(gen, B, N).cuda(), unit_ball(gen, B, S).cuda()
    feat = torch.randn(B, S, C, generator=gen).cuda().requires_grad_(True)
    d, i = ops.three_nn(xyz1, xyz2, 3)
    out = ops.three_interpolate(feat, d, i)
    rec = 1.0 / (d + 1e-8)
    w = rec / rec.sum(dim=2, keepdim=True)
    bi = torch.arange(B, device="cuda").view(B, 1, 1)
    ref = (feat[bi, i] * w.unsqueeze(-1)).sum(dim=2)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    gw = torch.randn_like(out)
    (out * gw).sum().backward()
    g_hip = feat.grad.clone()
    feat.grad = None
    (ref * gw).sum().backward()
    torch.testing.assert_close(g_hip, feat.grad, rtol=1e-5, atol=1e-5)


def test_edge_features_backward_matches_autograd(ops):
    gen = torch.Generator().manual_seed(4)
    B, N, D, k = 2, 130, 6, 7
    x = torch.randn(B, N, D, generator=gen).cuda().requires_grad_(True)
    idx = ops.knn(x.detach(), k)
    out = ops.edge_features(x, idx)
    bi = torch.arange(B, device="cuda").view(B, 1, 1)
    ctr = x.unsqueeze(2).expand(-1, -1, k, -1)
    ref = torch.cat([x[bi, idx] - ctr, ctr], dim=-1)
    assert torch.equal(out, ref)
    gw = torch.randn_like(out)
    (out * gw).sum().backward()
    g_hip = x.grad.clone()
    x.grad = None
    (ref * gw).sum().backward()
    torch.testing.assert_close(g_hip, x.grad, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ full-size properties (BASELINE cfg2/cfg3)
def test_full_size_properties_cfg2(ops):
    """B=16, N=16384: too big for the CPU oracle in seconds, so check size-independent properties."""
    gen = torch.Generator().manual_seed(0)
    B, N, S = 16, 16384, 1024
    xyz = unit_ball(gen, B, N).cuda()
    start = torch.randint(0, N, (B,), generator=gen).cuda()
    fps = ops.furthest_point_sample(xyz, S, start)
    assert torch.equal(fps[:, 0], start)
    assert all(len(torch.unique(fps[b])) == S for b in range(B))           # no point sampled twice
    # scene 0 against the oracle (the scenes are independent)
    assert np.array_equal(fps[0].cpu().numpy(),
                          orc.farthest_point_sample(xyz[:1].cpu().numpy(), S, start[:1].cpu().numpy())[0])
    new_xyz = ops.gather_rows(xyz, fps)
    ia, ib = ops.ball_query2([0.1, 0.2], [16, 32], xyz, new_xyz)
    bi = torch.arange(B, device="cuda").view(B, 1, 1)
    for idx, r in ((ia, 0.1), (ib, 0.2)):
        assert int(idx.min()) >= 0 and int(idx.max()) < N                  # every centroid finds itself
        d = ((xyz[bi, idx] - new_xyz.unsqueeze(2)) ** 2).sum(-1)
        assert float(d.max()) <= r * r * (1 + 1e-5)                         # members are inside the ball
        first, rest = idx[:, :, :1], idx[:, :, 1:]
        inc = (rest > idx[:, :, :-1]) | (rest == first)                     # ascending, then padding
        assert bool(inc.all())
    assert torch.equal(ops.ball_query(0.1, 16, xyz, new_xyz), ia)           # fused == single radius
    d3, i3 = ops.three_nn(xyz, new_xyz, 3)
    assert bool((d3[:, :, 1:] >= d3[:, :, :-1]).all())
    # a sampled point's nearest centroid is itself
    self_d, self_i = ops.three_nn(new_xyz, new_xyz, 3)
    assert torch.equal(self_i[:, :, 0], torch.arange(S, device="cuda").expand(B, S))
    od, oi = orc.three_nn(xyz[:1, :4096].cpu().numpy(), new_xyz[:1].cpu().numpy(), 3)
    assert np.array_equal(i3[0, :4096].cpu().numpy(), oi[0])


def test_full_size_properties_cfg3_knn(ops):
    gen = torch.Generator().manual_seed(1)
    B, N, k = 8, 8192, 20
    x = unit_ball(gen, B, N).cuda()
    idx = ops.knn(x, k)
    assert torch.equal(idx[:, :, 0], torch.arange(N, device="cuda").expand(B, N))  # self first
    bi = torch.arange(B, device="cuda").view(B, 1, 1)
    d = ((x[bi, idx] - x.unsqueeze(2)) ** 2).sum(-1)
    assert bool((d[:, :, 1:] >= d[:, :, :-1] - 1e-6).all())                        # sorted by distance
    assert np.array_equal(idx[0, :512].cpu().numpy(), orc.knn(x[:1].cpu().numpy(), k)[0, :512])


@pytest.mark.parametrize("kind,B,N,k", [("ball", 2, 16384, 32), ("ball", 3, 5000, 16), ("ball", 2, 1024, 20),
                                         ("flat", 2, 9000, 32), ("line", 2, 4096, 8), ("dup", 2, 6000, 20),
                                         ("grid", 2, 8000, 32), ("clusters", 2, 12345, 16), ("outlier", 2, 7000, 20),
                                         ("same", 1, 2048, 5), ("aniso", 2, 16000, 32), ("ball", 1, 16384, 1)])
def test_grid_knn_equals_all_pairs_kernel(ops, kind, B, N, k):
    """pcb_knn_xyz (grid search) must return exactly what pcb_knn returns on the coordinates -- same pd
    arithmetic, same tie rule, same order -- on clouds built to stress it: a plane, a line, repeated
    points, an exact lattice full of ties, tight clusters and a far outlier (crowded cells: the
    all-pairs fallback), one repeated point, anisotropic extents."""
    gen = torch.Generator().manual_seed(N + k)
    if kind == "flat":
        xyz = unit_ball(gen, B, N); xyz[..., 2] = -0.3
    elif kind == "line":
        xyz = torch.zeros(B, N, 3); xyz[..., 1] = torch.rand(B, N, generator=gen)
    elif kind == "dup":
        base = unit_ball(gen, B, N // 2); xyz = torch.cat([base, base], dim=1)
    elif kind == "grid":
        xyz = torch.randint(-16, 16, (B, N, 3), generator=gen).float() / 16.0
    elif kind == "clusters":
        centres = unit_ball(gen, B, 8)
        pick = torch.randint(0, 8, (B, N), generator=gen)
        xyz = torch.gather(centres, 1, pick.unsqueeze(-1).expand(-1, -1, 3)) + 0.01 * torch.randn(B, N, 3, generator=gen)
    elif kind == "outlier":
        xyz = unit_ball(gen, B, N); xyz[:, 0] = torch.tensor([300.0, -200.0, 50.0])
    elif kind == "same":
        xyz = torch.full((B, N, 3), 0.25)
    elif kind == "aniso":
        xyz = unit_ball(gen, B, N) * torch.tensor([1.0, 0.1, 0.03])
    else:
        xyz = unit_ball(gen, B, N)
    xyz = xyz.contiguous().cuda()
    try:
        ops.set_grid_knn(False)
        want = ops.knn(xyz, k)
        ops.set_grid_knn(True)
        got = ops.knn(xyz, k)
    finally:
        ops.set_grid_knn(True)
    assert torch.equal(got, want)
