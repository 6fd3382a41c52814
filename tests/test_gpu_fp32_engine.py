"""GPU: the fp32 row type of the fused MLP engine (csrc/gemm_f32.hip, the *_f32 row kernels) -- the
arithmetic the 1e-4 network parity tests (tests/test_gpu_modules.py, test_gpu_bridge.py) run on.

Every kernel is compared with an fp64 PyTorch evaluation of the same operator (forward, input /
weight / affine gradients, running statistics).  Bars are fp32's: 2e-5 of the largest magnitude
(the fp32 matrix core is an exact fma chain; what remains is summation order).
Also here: the slab-count contract of the statistics epilogues (include/pcb_hip.h) -- the caller's
count is the grid, whatever the concurrency hint says -- for both row types.
"""
import ctypes

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rm():
    assert torch.cuda.is_available()
    from pointcloud_bridge_amd import rowmlp
    rowmlp.set_precision("fp32")
    yield rowmlp
    rowmlp.set_precision("fp32")


def _act(u, act):
    return F.relu(u) if act == 1 else (F.leaky_relu(u, 0.2) if act == 2 else u)


def _near(a, b, tol=2e-5, atol=0.0):
    a, b = a.detach().double(), b.detach().double()
    scale = max(float(b.abs().max()), 1e-12)
    err = float((a - b).abs().max())
    if err > tol * scale + atol:
        print(f"_near failed: max |d| {err:.4g} vs max |b| {scale:.4g}, shape {tuple(b.shape)}")
    return err <= tol * scale + atol


def _near_q(a, b, tol, q=0.99):
    """Like _near, for gradients BEHIND a ReLU / max-pool: those are discontinuous, and a pre-activation
    within rounding distance of zero (or two pool candidates within rounding distance of each other)
    legitimately takes the other branch in fp32 than in fp64 -- which moves one whole row of an input
    gradient / one row of a weight gradient (measured: exactly one of 256 channels of one layer at
    R = 8192, every other channel at 5e-7; tools/f32_stack_debug.py).  So: the q-quantile of |a - b|
    must be within tol of the largest |b|, and nothing may be off by more than 50 %."""
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    d = (a - b).abs()
    scale = max(float(b.abs().max()), 1e-12)
    kth = max(1, int(d.numel() * q))
    quant = float(d.kthvalue(kth)[0]) if d.numel() > 1 else float(d.max())
    ok = quant <= tol * scale and float(d.max()) <= 0.5 * scale
    if not ok:
        print(f"_near_q failed: {q}-quantile |d| {quant:.4g}, max |d| {float(d.max()):.4g} vs max |b| {scale:.4g}")
    return ok


def _first_max_pool(h, pool):
    h3 = h.view(-1, pool, h.shape[1])
    eq = h3 == h3.max(dim=1, keepdim=True)[0]
    first = eq & (eq.cumsum(dim=1) == 1)
    return (h3 * first).sum(dim=1)


@pytest.mark.parametrize("R,K,widths,act,pool,perm", [
    (2048, 8, [64, 64, 128], 1, 16, 0), (1024, 260, [128, 128, 256], 1, 32, 0), (3000, 72, [256, 128], 1, 0, 0),
    (1280, 128, [64], 2, 20, 0), (1536, 20, [32, 32, 64], 1, 8, 16), (1100, 260, [256, 256], 1, 0, 0),
    (77 * 3, 12, [24], 1, 3, 0), (999, 4, [4, 8], 0, 0, 0), (640, 516, [256, 256, 512], 1, 32, 0),
    (2048, 264, [256, 128], 1, 0, 0), (4096, 264, [256, 128], 1, 0, 0), (4096, 264, [128, 128], 1, 0, 0),
    (8192, 64, [256, 128], 1, 0, 0), (4096, 264, [256, 128, 64], 1, 0, 0)])
def test_f32_stack_forward_backward_vs_torch_fp64(rm, R, K, widths, act, pool, perm):
    """mlp_rows in fp32 mode = GEMMs on the fp32 matrix cores with BatchNorm+activation on operand load,
    statistics / BatchNorm-backward sums from the epilogues, dy recomputed on load -- against the
    same stack of nn.Linear-like products + nn.BatchNorm1d in fp64."""
    torch.manual_seed(R + K)
    dev = "cuda"
    x = torch.randn(R, K, device=dev).requires_grad_(True)
    kin = K if not perm else perm + 3     # real input width when the rows carry padding
    dims = [kin] + widths
    convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip(dims[:-1], dims[1:])).to(dev)
    bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).to(dev).train()
    refs = nn.ModuleList(nn.BatchNorm1d(b) for b in widths).to(dev).double().train()
    with torch.no_grad():
        for bn, rf in zip(bns, refs):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            rf.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
        if perm:
            x.data[:, kin:] = 0          # padding columns as pcb_group_rows_f32 writes them
    out = rm.mlp_rows(convs, bns, x, act, pool, perm)
    assert out.dtype == torch.float32

    xr = x.detach().double().requires_grad_(True)
    h = xr[:, :kin]
    if perm:  # rows are [features | xyz]; the reference weight expects [xyz | features]
        h = torch.cat([h[:, perm:perm + 3], h[:, :perm]], dim=1)
    ws = []
    for conv, rf in zip(convs, refs):
        w = conv.weight.detach().view(conv.out_channels, -1).double().requires_grad_(True)
        ws.append(w)
        h = _act(rf(h @ w.t() + conv.bias.detach().double()), act)
    ref = _first_max_pool(h, pool) if pool else h
    assert out.shape == ref.shape
    assert _near(out, ref.detach())
    g = torch.randn_like(ref)
    (out * g.float()).sum().backward()
    (ref * g).sum().backward()
    def rel(a_, b_):
        return float((a_.detach().double() - b_.detach().double()).abs().max() / b_.detach().double().abs().max().clamp_min(1e-12))
    print("errors: dx %.2e" % rel(x.grad[:, :kin], xr.grad[:, :kin]),
          " ".join("| L%d dW %.2e dgamma %.2e dbeta %.2e" % (i, rel(c.weight.grad.view_as(w_), w_.grad), rel(b_.weight.grad, r_.weight.grad),
                                                           rel(b_.bias.grad, r_.bias.grad))
                   for i, (c, w_, b_, r_) in enumerate(zip(convs, ws, bns, refs))))
    assert _near_q(x.grad[:, :kin], xr.grad[:, :kin], 1e-4, 0.995)
    for conv, w, bn, rf in zip(convs, ws, bns, refs):
        assert _near_q(conv.weight.grad.view_as(w), w.grad, 1e-4, 0.98)
        # (a sum that is zero in exact arithmetic -- the shift of a layer in front of another BatchNorm --
        # comes out as rounding noise of its terms: absolute floor)
        assert _near_q(bn.weight.grad + 0, rf.weight.grad, 1e-4, 0.98) or _near(bn.weight.grad, rf.weight.grad, 1e-4, atol=1e-3)
        assert _near_q(bn.bias.grad + 0, rf.bias.grad, 1e-4, 0.98) or _near(bn.bias.grad, rf.bias.grad, 1e-4, atol=1e-3)
        assert float(conv.bias.grad.abs().max()) == 0.0  # exactly zero under batch statistics
        assert _near(bn.running_mean, rf.running_mean, 1e-5)
        assert _near(bn.running_var, rf.running_var, 1e-5)
        assert int(bn.num_batches_tracked) == 1


def test_f32_eval_mode_uses_running_stats_and_bias(rm):
    torch.manual_seed(0)
    dev = "cuda"
    R, K, C = 1024, 32, 64
    conv = nn.Conv1d(K, C, 1).to(dev)
    bn = nn.BatchNorm1d(C).to(dev)
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    bn.eval()
    x = torch.randn(R, K, device=dev, requires_grad=True)
    out = rm.conv_bn_act(conv, bn, x, rm.ACT_RELU)
    xr = x.detach().double().requires_grad_(True)
    import copy
    bn64 = copy.deepcopy(bn).double().eval()
    ref = F.relu(bn64(F.linear(xr, conv.weight.detach().view(C, K).double(), conv.bias.detach().double())))
    assert _near(out, ref.detach())
    out.sum().backward()
    ref.sum().backward()
    assert _near(x.grad, xr.grad, 1e-4)
    assert conv.bias.grad is not None and float(conv.bias.grad.abs().max()) > 0  # no cancellation in eval mode


@pytest.mark.parametrize("R,K,n,gap", [(4096, 128, 5, 0), (1000, 16, 128, 0), (2048, 64, 259, 3), (512, 260, 13, 5)])
def test_f32_conv_rows_bias_and_gradients(rm, R, K, n, gap):
    """conv_rows (no BatchNorm): x W^T + b with the bias in the GEMM epilogue; optional gap layout."""
    torch.manual_seed(n)
    dev = "cuda"
    conv = nn.Conv1d(K, n, 1).to(dev)
    x = torch.randn(R, K, device=dev, requires_grad=True)
    y = rm.conv_rows(conv, x, out_gap=gap)
    xr = x.detach().double().requires_grad_(True)
    ref = F.linear(xr, conv.weight.detach().view(n, K).double(), conv.bias.detach().double())
    if gap:
        dp = (gap + 3) // 4 * 4
        cols = torch.cat([torch.arange(gap), dp + torch.arange(n - gap)]).to(dev)
        assert y.shape[1] % 4 == 0
        pad = torch.ones(y.shape[1], dtype=torch.bool, device=dev)
        pad[cols] = False
        assert not pad.any() or float(y[:, pad].abs().max()) == 0.0
        got = y[:, cols]
    else:
        got = y
    assert _near(got, ref.detach())
    g = torch.randn(R, n, device=dev)
    (got * g).sum().backward()
    (ref * g.double()).sum().backward()
    assert _near(x.grad, xr.grad, 1e-4)
    assert _near(conv.weight.grad.view(n, K), g.double().t() @ x.detach().double(), 1e-4)
    assert _near(conv.bias.grad, g.double().sum(0), 1e-4)


@pytest.mark.parametrize("perm,k,kp", [(0, 6, 8), (0, 64, 64), (3, 6, 8), (128, 131, 132), (-3, 259, 260), (-5, 13, 16)])
def test_f32_prep_weights_and_wgrad_layout(rm, perm, k, kp):
    """pcb_prep_weights_f32 == padded_weight_from(quantum 4) (+ transpose); pcb_gemm_tn_f32's
    out_cols/out_perm output == the unpadded columns of the padded result (bitwise)."""
    from pointcloud_bridge_amd import _lib
    from pointcloud_bridge_amd.ops import _launch
    from tests.test_gpu_bf16 import _unpad_weight_grad
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(11)
    C, R = 64, 4096
    w = torch.randn(C, k, 1, 1, generator=g).to(dev)
    ref = rm.padded_weight_from(w, kp, perm, quantum=4)
    wp = torch.empty(C, kp, dtype=torch.float32, device=dev)
    wt = torch.empty(kp, C, dtype=torch.float32, device=dev)
    desc = (ctypes.c_longlong * 8)(w.data_ptr(), wp.data_ptr(), wt.data_ptr(), C, k, kp, perm, 0)
    _launch("pcb_prep_weights_f32", 0, 1, desc)
    assert torch.equal(wp, ref) and torch.equal(wt, ref.t().contiguous())

    dz = torch.randn(R, C, generator=g).to(dev)
    x = torch.randn(R, kp, generator=g).to(dev)
    lib = _lib.load()
    ws = torch.empty(lib.pcb_gemm_tn_workspace(R, C, kp), dtype=torch.float32, device=dev)
    full = torch.empty(C, kp, dtype=torch.float32, device=dev)
    real = torch.full((C, k), float("nan"), dtype=torch.float32, device=dev)
    for out, cols, pm in ((full, 0, 0), (real, k, perm)):
        _launch("pcb_gemm_tn_f32", 0, 0, dz.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, x.data_ptr(), 0, 0, 0, R, C, kp,
                ws.data_ptr(), out.data_ptr(), cols, pm)
    assert torch.equal(real, _unpad_weight_grad(full, k, perm, 4).contiguous())
    assert _near(full, dz.double().t() @ x.double(), 2e-5)


@pytest.mark.parametrize("D1,C,k", [(3, 64, 4), (0, 32, 3), (16, 256, 4), (5, 8, 3)])
def test_f32_interpolate_concat_forward_backward(rm, D1, C, k):
    """interpolate_concat in fp32 rows == cat([skip, three_interpolate]) of the fp32 operators (bitwise
    forward: same rounding order), gradients against autograd of the plain composition."""
    from pointcloud_bridge_amd import ops
    torch.manual_seed(C + k)
    dev = "cuda"
    B, N, S = 2, 500, 64
    xyz1, xyz2 = torch.rand(B, N, 3, device=dev), torch.rand(B, S, 3, device=dev)
    d2, idx = ops.three_nn(xyz1, xyz2, k)
    feat = torch.randn(B, S, C, device=dev, requires_grad=True)
    skip = torch.randn(B * N, D1, device=dev, requires_grad=True) if D1 else None
    rows, perm = rm.interpolate_concat(skip, feat, d2, idx)
    dp = (D1 + 3) // 4 * 4
    assert rows.shape == (B * N, dp + C) and perm == (-D1 if D1 % 4 else 0)
    ref_i = ops.three_interpolate(feat.detach().clone().requires_grad_(True), d2, idx)
    assert torch.equal(rows[:, dp:], ref_i.reshape(B * N, C))
    if D1:
        assert torch.equal(rows[:, :D1], skip) and (dp == D1 or float(rows[:, D1:dp].abs().max()) == 0.0)
    g = torch.randn_like(rows)
    (rows * g).sum().backward()
    f2 = feat.detach().clone().requires_grad_(True)
    (ops.three_interpolate(f2, d2, idx).reshape(B * N, C) * g[:, dp:]).sum().backward()
    assert _near(feat.grad, f2.grad, 1e-5)
    if D1:
        assert torch.equal(skip.grad, g[:, :D1])


def test_f32_group_rows_layout_and_backward(rm):
    """group_rows in fp32 mode: [features | xyz_j - c_s | 0] rows, exact values; scatter-add backward."""
    from pointcloud_bridge_amd import ops
    torch.manual_seed(4)
    dev = "cuda"
    B, N, S, ns, C = 2, 300, 40, 8, 14
    xyz = torch.rand(B, N, 3, device=dev)
    new_xyz = xyz[:, :S].contiguous()
    idx = torch.randint(0, N, (B, S, ns), device=dev)
    feat = torch.randn(B, N, C, device=dev, requires_grad=True)
    rows, perm = rm.group_rows(xyz, new_xyz, feat, idx)
    kp = (C + 3 + 3) // 4 * 4
    assert rows.shape == (B * S * ns, kp) and perm == C and rows.dtype == torch.float32
    ref = ops.group_points(xyz, new_xyz, feat.detach(), idx).view(B * S * ns, 3 + C)  # coordinates first
    assert torch.equal(rows[:, :C], ref[:, 3:]) and torch.equal(rows[:, C:C + 3], ref[:, :3])
    assert kp == C + 3 or float(rows[:, C + 3:].abs().max()) == 0.0
    g = torch.randn_like(rows)
    (rows * g).sum().backward()
    want = torch.zeros(B * N, C, device=dev, dtype=torch.float64)
    src = (idx + torch.arange(B, device=dev).view(B, 1, 1) * N).reshape(-1)
    want.index_add_(0, src, g[:, :C].double())
    assert _near(feat.grad.view(B * N, C), want, 1e-5)


def test_f32_gate_and_bn_act_rows(rm):
    torch.manual_seed(2)
    dev = "cuda"
    x = torch.randn(1000, 64, device=dev, requires_grad=True)
    a = torch.randn(1000, 64, device=dev, requires_grad=True)
    out = rm.gate_rows(x, a)
    xr, ar = x.detach().double().requires_grad_(True), a.detach().double().requires_grad_(True)
    ref = xr * torch.sigmoid(ar)
    assert _near(out, ref.detach(), 1e-6)
    g = torch.randn_like(out)
    (out * g).sum().backward()
    (ref * g.double()).sum().backward()
    assert _near(x.grad, xr.grad, 1e-6) and _near(a.grad, ar.grad, 1e-5)

    bn = nn.BatchNorm1d(320).to(dev).train()
    rf = nn.BatchNorm1d(320).to(dev).double().train()
    y = torch.randn(4096, 320, device=dev, requires_grad=True) * 2 + 0.5
    y.retain_grad()
    out = rm.bn_act_rows(bn, y, rm.ACT_LEAKY)
    yr = y.detach().double().requires_grad_(True)
    ref = F.leaky_relu(rf(yr), 0.2)
    assert _near(out, ref.detach())
    g = torch.randn_like(out)
    (out * g).sum().backward()
    (ref * g.double()).sum().backward()
    assert _near(y.grad, yr.grad, 1e-4)
    assert _near(bn.weight.grad, rf.weight.grad, 1e-4) and _near(bn.bias.grad, rf.bias.grad, 1e-4)
    assert _near(bn.running_var, rf.running_var, 1e-5) and int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("sfx,dtype,q", [("bf16", torch.bfloat16, 8), ("f32", torch.float32, 4)])
def test_statistics_slab_count_is_the_callers(sfx, dtype, q):
    """include/pcb_hip.h: a gemm_nt launch given `sums` runs EXACTLY nparts workgroups along its row
    axis and writes exactly nparts slabs -- also when nparts exceeds the row tiles (zero slabs), also
    when the concurrency hint changes between the query and the launch.  Bytes behind the caller's
    slabs stay untouched; an invalid count is refused."""
    from pointcloud_bridge_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    torch.manual_seed(9)
    R, K, N = 1000, 32, 64                      # 8 row tiles
    x = torch.randn(R, K, device=dev).to(dtype)
    w = torch.randn(N, K, device=dev).to(dtype)
    out = torch.empty(R, N, dtype=dtype, device=dev)
    gemm = getattr(lib, "pcb_gemm_nt_" + sfx)
    st = torch.cuda.current_stream().cuda_stream
    yref = (x.float() @ w.float().t()).to(dtype).float()
    try:
        for nparts in (1, 3, 8, 20):
            lib.pcb_set_concurrency_hint(0)
            assert lib.pcb_gemm_nt_partials(0, R, N) == 8
            lib.pcb_set_concurrency_hint(48)    # changes the library's own preference after the "query"
            guard = 4
            sums = torch.full((nparts + guard, 2, N), 12345.0, device=dev)
            assert gemm(0, x.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 0, 0, w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr(),
                        nparts, st) == 0
            torch.cuda.synchronize()
            assert float((sums[nparts:] - 12345.0).abs().max()) == 0.0, "wrote past the caller's slabs"
            tot = sums[:nparts].sum(0)
            assert torch.allclose(tot[0], yref.sum(0), rtol=1e-3, atol=1e-2)
            assert torch.allclose(tot[1], (yref * yref).sum(0), rtol=1e-3, atol=1e-2)
            assert torch.allclose(out.float(), yref, rtol=2e-2 if q == 8 else 1e-5, atol=1e-2 if q == 8 else 1e-4)
        for bad in (0, -1, 769):
            assert gemm(0, x.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 0, 0, w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr(),
                        bad, st) == -1
    finally:
        lib.pcb_set_concurrency_hint(0)


def test_fp32_mode_runs_no_library_gemm_or_aten_batchnorm(rm):
    """The parity mode's pointwise MLPs are own kernels: a PointNet++ MSG step in fp32 mode launches no
    hipBLASLt / rocBLAS GEMM (Cijk_*) and no ATen batch_norm kernel."""
    from torch.profiler import ProfilerActivity, profile
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    torch.manual_seed(0)
    dev = "cuda"
    model = PointNet2MSG(5).to(dev).train()
    B, N = 2, 2048
    xyz = torch.rand(B, N, 3, device=dev)
    col = torch.rand(B, N, 3, device=dev)
    lab = torch.randint(0, 5, (B, N), device=dev)
    F.cross_entropy(model(xyz, col), lab).backward()  # warm-up
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model.zero_grad(set_to_none=True)
        F.cross_entropy(model(xyz, col), lab).backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    bad = [n for n in names if n.startswith("Cijk_") or "batch_norm" in n.lower() or n in ("aten::mm", "aten::addmm", "aten::linear", "aten::bmm")]
    assert not bad, bad
    assert any("gemm_nt_f32_kernel" in n for n in names) and any("gemm_tn_f32_kernel" in n for n in names)
