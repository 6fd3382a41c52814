"""GPU: the reproducible mode (round 3) -- every backward pass of the package in a fixed summation order.

The reference's gathers differentiate into index_put_(accumulate=True) (index_points models/pointnet2_utils.py:17-39,
the grouping :51-58 / :342-349, DGCNN.get_graph_feature models/DGCNN.py:90-107).  The throughput path adds with fp32
atomics; ops.set_deterministic(True) (or PCB_DETERMINISTIC=1, or torch.use_deterministic_algorithms) routes them
through segment sums over an inverted index (csrc/segsum.hip), the gathered first layers through
pcb_scatter_dy_bf16(det) + pcb_scatter_dy_csr_bf16, and the remaining atomically accumulated statistics through slabs.
Checked: the kernels against index_add_ evaluations, and whole training steps -- two runs, bit-identical gradients.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def det():
    from pointcloud_bridge_amd import ops
    old = ops.set_deterministic(True)
    yield ops
    ops.set_deterministic(old)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,ld,col0", [(64, 64, 0), (3, 8, 2), (200, 264, 8)])
def test_segment_sum_against_index_add(det, dtype, C, ld, col0):
    ops = det
    g = torch.Generator().manual_seed(C)
    B, N, M = 3, 50, 700
    idx = torch.randint(0, N, (B, M), generator=g).cuda()
    idx[:, :40] = 7                                  # a long segment; most targets of scene 2 stay empty
    idx[2] = idx[2] % 5
    rows = torch.randn(B * M, ld, generator=g).cuda().to(dtype)
    order, offsets = ops.det_index(idx, N)
    assert order.dtype == torch.int32 and offsets.shape == (B * N + 1,) and int(offsets[-1]) == B * M
    seg = torch.repeat_interleave(torch.arange(B * N, device="cuda"), offsets[1:] - offsets[:-1])
    tgt = (idx + torch.arange(B, device="cuda").view(B, 1) * N).reshape(-1)
    assert torch.equal(tgt[order.long()], seg)       # sorted by target ...
    assert bool(((order[1:] > order[:-1]) | (seg[1:] != seg[:-1])).all())   # ... ascending source rows inside a segment
    out = torch.full((B * N, C), float("nan"), device="cuda")
    ops.segment_sum(rows, col0, C, order, offsets, out)
    ref = torch.zeros(B * N, C, dtype=torch.float64, device="cuda").index_add_(0, tgt, rows[:, col0:col0 + C].double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    base = torch.randn(B * N, C, device="cuda")
    acc = base.clone()
    ops.segment_sum(rows, col0, C, order, offsets, acc, accumulate=True)
    torch.testing.assert_close(acc.double(), ref + base.double(), rtol=1e-5, atol=1e-5)
    again = torch.empty_like(out)
    ops.segment_sum(rows, col0, C, order, offsets, again)
    assert torch.equal(out, again)


@pytest.mark.parametrize("pooled,with_wx,C", [(0, 0, 64), (1, 1, 64), (0, 1, 192), (1, 0, 520)])
def test_scatter_dy_reproducible_form(det, pooled, with_wx, C):
    """pcb_scatter_dy_bf16(det = 1) + pcb_scatter_dy_csr_bf16 against the atomic form: same du / dv / dWx up to the fp32
    summation order, and bit-identical when run twice."""
    from pointcloud_bridge_amd import _lib
    from pointcloud_bridge_amd.ops import _launch
    ops = det
    lib = _lib.load()
    g = torch.Generator().manual_seed(21)
    B, N, S, ns = 2, 96, 40, 8
    R = B * S * ns
    dev = torch.device("cuda")
    xyz, ctr = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, S, 3, generator=g).to(dev)
    idx = torch.randint(0, N, (B, S, ns), generator=g).to(dev)
    idx[:, :, ns // 2:] = idx[:, :, :1]
    y = torch.randn(R, C, generator=g).to(dev).to(torch.bfloat16)
    scale, shift = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev) * 0.3
    p, q = torch.randn(C, generator=g).to(dev) * 0.1, torch.randn(C, generator=g).to(dev) * 0.1
    if pooled:
        dout = torch.randn(B * S, C, generator=g).to(dev)
        arg = torch.randint(0, ns, (B * S, C), generator=g).to(dev).to(torch.uint8)
        dz = None
    else:
        dz = torch.randn(R, C, generator=g).to(dev).to(torch.bfloat16)
        dout = arg = None

    def run(det_flag):
        slabs = lib.pcb_scatter_dy_slabs(B, S, C, det_flag)
        assert slabs == 33 if not det_flag else slabs > 1
        du = torch.zeros(B * N, C, device=dev) if not det_flag else torch.full((B * N, C), float("nan"), device=dev)
        dv = torch.empty(B * S, C, device=dev)
        dwx = torch.zeros(slabs, C, 3, device=dev) if with_wx else None
        _launch("pcb_scatter_dy_bf16", 0, pooled, 0 if dz is None else dz.data_ptr(), y.data_ptr(), scale.data_ptr(),
                shift.data_ptr(), p.data_ptr(), q.data_ptr(), 0 if dout is None else dout.data_ptr(),
                0 if arg is None else arg.data_ptr(), 1, idx.data_ptr(), B, N, S, ns, C, xyz.data_ptr(), ctr.data_ptr(),
                0 if det_flag else du.data_ptr(), dv.data_ptr(), 0 if dwx is None else dwx.data_ptr(), det_flag)
        if det_flag:
            order, offsets = ops.det_index(idx, N)
            _launch("pcb_scatter_dy_csr_bf16", 0, pooled, 0 if dz is None else dz.data_ptr(), y.data_ptr(), scale.data_ptr(),
                    shift.data_ptr(), p.data_ptr(), q.data_ptr(), 0 if dout is None else dout.data_ptr(),
                    0 if arg is None else arg.data_ptr(), 1, ns, C, order.data_ptr(), offsets.data_ptr(), B * N, du.data_ptr())
        return du, dv, None if dwx is None else dwx[0].clone()

    a, b, c = run(0), run(1), run(1)
    torch.testing.assert_close(b[0], a[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b[1], a[1], rtol=1e-4, atol=1e-4)
    if with_wx:
        torch.testing.assert_close(b[2], a[2], rtol=1e-4, atol=1e-3)
    for u, v in zip(b, c):
        assert u is None or torch.equal(u, v)


def _ops_backward_cases():
    from pointcloud_bridge_amd import ops
    g = torch.Generator().manual_seed(3)
    B, N, S, ns, C = 2, 300, 64, 16, 12
    xyz = torch.rand(B, N, 3, generator=g).cuda()
    new_xyz = xyz[:, :S].contiguous()
    idx = torch.randint(0, N, (B, S, ns), generator=g).cuda()
    feat = torch.randn(B, N, C, generator=g).cuda()
    knn = torch.randint(0, N, (B, N, 8), generator=g).cuda()
    d2, nn3 = ops.three_nn(xyz, new_xyz, 3)
    coarse = torch.randn(B, S, C, generator=g).cuda()
    return {
        "gather_rows": (lambda f: ops.gather_rows(f, idx), feat),
        "group_points": (lambda f: ops.group_points(xyz, new_xyz, f, idx), feat),
        "edge_features": (lambda f: ops.edge_features(f, knn), feat),
        "three_interpolate": (lambda f: ops.three_interpolate(f, d2, nn3), coarse),
    }


@pytest.mark.parametrize("name", ["gather_rows", "group_points", "edge_features", "three_interpolate"])
def test_gather_backwards_reproducible_and_equal_to_the_atomic_form(name):
    from pointcloud_bridge_amd import ops
    fn, leaf = _ops_backward_cases()[name]

    def grad(det_flag):
        old = ops.set_deterministic(det_flag)
        try:
            x = leaf.clone().requires_grad_(True)
            out = fn(x)
            w = torch.linspace(-1, 1, out.numel(), device="cuda").view_as(out)
            (out * w).sum().backward()
            return x.grad.clone()
        finally:
            ops.set_deterministic(old)

    a, b, c = grad(False), grad(True), grad(True)
    torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5)
    assert torch.equal(b, c)


def _step_gradients(model_name, precision, det_flag, steps=2):
    """Flat gradient of every step of `steps` training steps from a fixed initial state (model seed, data seed, FPS seed)."""
    import bench
    from pointcloud_bridge_amd import ops, rowmlp
    old = ops.set_deterministic(det_flag)
    try:
        torch.manual_seed(42)
        model, cdim = bench.build_model(model_name)
        model = model.cuda().train()
        B, N = (2, 2048) if model_name != "dgcnn" else (2, 1024)
        xyz, colors, labels = bench.synthetic_batch(B, N, 5, "cuda")
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        torch.manual_seed(9)            # CPU generator: FPS start indices; CUDA generator: dropout masks
        torch.cuda.manual_seed(9)
        grads, losses = [], []
        with rowmlp.precision(precision):
            for _ in range(steps):
                opt.zero_grad(set_to_none=True)
                loss = bench.loss_fn(model(xyz, colors), labels, cdim)
                loss.backward()
                grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).clone())
                losses.append(float(loss))
                opt.step()
        return grads, losses
    finally:
        ops.set_deterministic(old)


@pytest.mark.parametrize("model_name,precision", [("pn2_msg", "bf16"), ("pn2_msg", "fp32"), ("pn2_ssg", "bf16"),
                                                  ("dgcnn", "bf16"), ("dgcnn", "fp32")])
def test_training_steps_are_bit_reproducible(model_name, precision):
    """Two runs of the same two training steps (the second step starts from weights the first one's gradients moved):
    bit-identical losses and gradients in the reproducible mode; the same gradients as the atomic mode up to the
    reordering noise of its float atomics."""
    ga, la = _step_gradients(model_name, precision, True)
    gb, lb = _step_gradients(model_name, precision, True)
    assert la == lb
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)
    gc, lc = _step_gradients(model_name, precision, False)
    rel = float((gc[0] - ga[0]).norm() / ga[0].norm())
    print(model_name, precision, "first-step gradient, atomic vs reproducible mode: relative L2", rel)
    # (bf16 rows: an atomic order that changes one bf16 rounding of a scattered gradient moves things at the 1e-2 level;
    # DGCNN feeds such differences back through its feature-space kNN graphs in the second step only)
    assert rel < ((5e-3 if model_name == "dgcnn" else 1e-4) if precision == "fp32" else 5e-2)
