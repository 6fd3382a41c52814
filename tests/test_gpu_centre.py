"""GPU: centred storage of pre-BatchNorm rows in the bf16 engine (round 3).

The reference keeps every activation in fp32 (models/pointnet2_utils.py:149-154, :207-209, :353-356); the bf16 engine
stores only y = x W^T per layer, and a bf16 y carries an absolute error of 2^-9 |y| that train-mode BatchNorm divides by
std(y).  BatchNorm is invariant to a per-channel constant in front of it, so the GEMM epilogue subtracts one close to
the batch mean before rounding (pcb_gemm_nt_stats_bf16, pcb_gather_add_bf16, pcb_gemm_nt_stats_add_bf16) and
pcb_bn_finalize_centred keeps the books.  Checked here: the entry points against their definitions (ragged row counts:
padding rows must stay out of the statistics), the three finalize modes, and at stack level that NOTHING a caller can
observe depends on the centre except the rounding error, which shrinks when |mean| >> std.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("R,N,K,pro", [(900, 512, 384, 0), (900, 64, 8, 0), (3000, 256, 72, 0), (900, 320, 512, 1),
                                       (1000, 128, 640, 1), (128 * 5, 264, 264, 0)])
def test_centred_gemm_entry_point(R, N, K, pro):
    """out = bf16(A' W^T - centre) and the statistics slabs of exactly those rows; slabs of idle workgroups read zero;
    K > 512 with the BatchNorm prologue takes the four-wave kernel, everything else the eight-wave one."""
    from pointcloud_bridge_amd import _lib
    L = _lib.load()
    torch.manual_seed(R + N)
    a = torch.randn(R, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.2).to(torch.bfloat16)
    scale, shift = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.3
    centre = torch.randn(N, device="cuda") * 2
    x = a.float()
    if pro:
        x = torch.relu(x * scale + shift).to(torch.bfloat16).float()
    prod = x @ w.float().t()
    for c in (centre, None):
        out = torch.empty(R, N, dtype=torch.bfloat16, device="cuda")
        nparts = 13
        sums = torch.full((nparts, 2, N), float("nan"), device="cuda")
        assert L.pcb_gemm_nt_stats_bf16(pro, a.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, w.data_ptr(), R, N, K,
                                        out.data_ptr(), sums.data_ptr(), nparts, 0 if c is None else c.data_ptr(), _stream()) == 0
        ref = prod if c is None else prod - c
        assert float((out.float() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
        tot, o = sums.double().sum(0), out.double()
        assert not torch.isnan(tot).any()
        assert float(((tot[0] - o.sum(0)).abs() / (o.abs().sum(0) + 1e-9)).max()) < 1e-5
        assert float(((tot[1] - (o * o).sum(0)).abs() / ((o * o).sum(0) + 1e-9)).max()) < 1e-5


def test_finalize_modes():
    """pcb_bn_finalize_centred: cmode 1 (training) = the uncentred call's running statistics and invstd, constants in the
    centred frame, centre moved to the batch mean; cmode 2 (probe) moves the centre and touches nothing else; eval
    cmode 1 writes running_mean - bias and a zero mean."""
    from pointcloud_bridge_amd import _lib
    L = _lib.load()
    torch.manual_seed(3)
    R, C = 5000, 96
    y = torch.randn(R, C, device="cuda") * torch.rand(C, device="cuda") + torch.randn(C, device="cuda") * 5
    c0 = y.mean(0) + torch.randn(C, device="cuda") * 0.1
    gamma, beta, bias = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda"), torch.randn(C, device="cuda")

    def call(rows, centre, cmode, training, rm, rv, nbt):
        sums = torch.stack([rows.sum(0), (rows * rows).sum(0)]).contiguous()
        outs = [torch.full((C,), float("nan"), device="cuda") for _ in range(4)]
        assert L.pcb_bn_finalize_centred(sums.data_ptr(), 1, R, 0, C, gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(),
                                         rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, training, *[o.data_ptr() for o in outs],
                                         nbt.data_ptr(), 0 if centre is None else centre.data_ptr(), cmode, _stream()) == 0
        return outs

    rm_a, rv_a, nbt_a = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    rm_b, rv_b, nbt_b = rm_a.clone(), rv_a.clone(), nbt_a.clone()
    sc_a, sh_a, mu_a, is_a = call(y, None, 0, 1, rm_a, rv_a, nbt_a)
    centre = c0.clone()
    sc_b, sh_b, mu_b, is_b = call(y - c0, centre, 1, 1, rm_b, rv_b, nbt_b)
    # (the uncentred call forms E[y^2] - mean^2 from fp32 sums of values with |mean| up to 15 std: it is the LESS accurate
    # of the two; the truth is the fp64 evaluation)
    var = y.double().var(0, unbiased=False)
    istd = (1.0 / torch.sqrt(var + 1e-5)).float()
    torch.testing.assert_close(rm_b, rm_a, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rv_b, (0.9 + 0.1 * var * R / (R - 1)).float(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rv_a, rv_b, rtol=5e-3, atol=1e-5)
    torch.testing.assert_close(is_b, istd, rtol=1e-4, atol=0)
    torch.testing.assert_close(is_a, istd, rtol=5e-3, atol=0)
    torch.testing.assert_close(sc_b, gamma * istd, rtol=1e-4, atol=0)
    torch.testing.assert_close(mu_b + c0, mu_a, rtol=1e-5, atol=1e-5)
    # next step's centre = this batch's mean where it had drifted by more than std/4, untouched elsewhere
    moved = (y.mean(0) - c0).abs() * istd > 0.25
    assert 5 < int(moved.sum()) < C - 5
    torch.testing.assert_close(centre[moved], y.mean(0)[moved], rtol=1e-5, atol=1e-5)
    assert torch.equal(centre[~moved], c0[~moved])
    # same normalised output from either frame
    torch.testing.assert_close((y - c0) * sc_b + sh_b, y * sc_a + sh_a, rtol=5e-3, atol=5e-3)
    assert int(nbt_a) == 1 and int(nbt_b) == 1
    # probe: only the centre moves
    centre = torch.zeros(C, device="cuda")
    rm_c, rv_c, nbt_c = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    outs = call(y, centre, 2, 1, rm_c, rv_c, nbt_c)
    torch.testing.assert_close(centre, y.mean(0), rtol=1e-5, atol=1e-5)
    assert int(nbt_c) == 0 and float(rm_c.abs().max()) == 0.0 and float((rv_c - 1).abs().max()) == 0.0
    assert all(torch.isnan(o).all() for o in outs)
    # eval
    centre = torch.full((C,), float("nan"), device="cuda")
    sc_e, sh_e, mu_e, is_e = call(y, centre, 1, 0, rm_a, rv_a, nbt_a)
    torch.testing.assert_close(centre, rm_a - bias)
    assert float(mu_e.abs().max()) == 0.0
    torch.testing.assert_close(sh_e, beta)
    torch.testing.assert_close(sc_e, gamma / torch.sqrt(rv_a + 1e-5))


def _stack(widths, K, offset):
    torch.manual_seed(11)
    dims = [K] + widths
    convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip(dims[:-1], dims[1:])).cuda()
    bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).cuda().train()
    with torch.no_grad():
        for conv in convs:
            conv.bias.normal_(0, offset)     # the reference adds it in front of BatchNorm; it only reaches running_mean
    return convs, bns


@pytest.mark.parametrize("R,K,widths,pool", [(4000, 24, [64, 64, 128], 16), (1100, 264, [256, 256], 0), (999, 72, [136], 0)])
def test_centred_stack_is_closer_to_fp32_when_the_mean_dominates(R, K, widths, pool):
    """A stack whose inputs carry a large common offset (|mean y| >> std y): outputs, running statistics and gradients of
    the centred bf16 engine against an fp64 evaluation -- never worse than uncentred storage, and several times better
    on the outputs; a second training step (centre = previous batch mean, no probe) keeps that."""
    from pointcloud_bridge_amd import rowmlp
    torch.manual_seed(R)
    x = (torch.randn(R, K, device="cuda") * 0.25 + 3.0).to(torch.bfloat16)    # offset rows: y = W x has |mean| >> std
    g = torch.randn(R // pool if pool else R, widths[-1], device="cuda")

    def reference():
        convs, bns = _stack(widths, K, 0.5)
        h = x.double()
        ps = []
        for conv, bn in zip(convs, bns):
            w = conv.weight.detach().view(conv.out_channels, -1).to(torch.bfloat16).double().requires_grad_(True)
            ps.append(w)
            z = h @ w.t()
            h = F.relu((z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + bn.eps) * bn.weight.double() + bn.bias.double())
        if pool:
            h = h.view(-1, pool, h.shape[1]).max(dim=1)[0]
        (h * g.double()).sum().backward()
        return h.detach(), [p.grad for p in ps]

    ref, ref_grads = reference()

    def run(centring, steps):
        convs, bns = _stack(widths, K, 0.5)
        old = rowmlp.set_centring(centring)
        try:
            with rowmlp.precision("bf16"):
                for _ in range(steps):
                    for p in convs.parameters():
                        p.grad = None
                    out = rowmlp.mlp_rows(convs, bns, x, rowmlp.ACT_RELU, pool)
                    (out.float() * g).sum().backward()
        finally:
            rowmlp.set_centring(old)
        e_out = float((out.double() - ref).abs().mean() / ref.abs().mean())
        e_grad = max(float((c.weight.grad.view_as(r).double() - r).abs().mean() / r.abs().mean()) for c, r in zip(convs, ref_grads))
        return e_out, e_grad, bns

    e0, g0, _ = run(False, 1)
    e1, g1, bns1 = run(True, 1)
    e2, g2, bns2 = run(True, 2)
    print(f"output error vs fp64: uncentred {e0:.3e}, centred {e1:.3e} (probe), {e2:.3e} (second step); weight gradients {g0:.3e} {g1:.3e} {g2:.3e}")
    assert e1 < 0.5 * e0 and e2 < 0.5 * e0
    assert g1 < 1.2 * g0 and g2 < 1.2 * g0
    # running statistics as nn.BatchNorm keeps them (bias included), num_batches_tracked once per call
    convs, bns = _stack(widths, K, 0.5)
    h = x.float()
    for conv, bn, b1, b2 in zip(convs, bns, bns1, bns2):
        z = h @ conv.weight.detach().view(conv.out_channels, -1).to(torch.bfloat16).float().t() + conv.bias.detach()
        h = F.relu(bn(z.t().reshape(1, -1, z.shape[0], 1)).reshape(z.shape[1], z.shape[0]).t())
        torch.testing.assert_close(b1.running_mean, bn.running_mean, rtol=2e-2, atol=2e-3)
        torch.testing.assert_close(b1.running_var, bn.running_var, rtol=3e-2, atol=3e-3)
        assert int(b1.num_batches_tracked) == 1 and int(b2.num_batches_tracked) == 2
        # the centre a layer keeps is its last batch mean (without the bias)
        torch.testing.assert_close(b1._pcb_centre, (z - conv.bias.detach()).mean(0), rtol=2e-2, atol=2e-2)


def test_eval_mode_is_centred_on_the_running_mean():
    """Eval mode stores y - (running_mean - bias): with running statistics far from zero the bf16 engine's eval output is
    closer to the fp32 evaluation than uncentred storage, and the cached-operand path (second no-grad call) returns the
    same bits."""
    from pointcloud_bridge_amd import rowmlp
    torch.manual_seed(0)
    R, K, C = 2048, 32, 64
    x = torch.randn(R, K, device="cuda") * 0.2 + 2.0
    conv = nn.Conv1d(K, C, 1).cuda()
    bn = nn.BatchNorm1d(C).cuda()
    w = conv.weight.view(C, K).to(torch.bfloat16).float()
    with torch.no_grad():
        z = x.to(torch.bfloat16).float() @ w.t() + conv.bias
        bn.running_mean.copy_(z.mean(0))
        bn.running_var.copy_(z.var(0))
    bn.eval()
    ref = F.relu(bn(z))
    errs = {}
    for centring in (False, True):
        old = rowmlp.set_centring(centring)
        try:
            with rowmlp.precision("bf16"), torch.no_grad():
                out = rowmlp.conv_bn_act(conv, bn, x, rowmlp.ACT_RELU)
                again = rowmlp.conv_bn_act(conv, bn, x, rowmlp.ACT_RELU)
        finally:
            rowmlp.set_centring(old)
        assert torch.equal(out, again)
        errs[centring] = float((out.float() - ref).abs().mean() / ref.abs().mean())
    print("eval error vs fp32:", errs)
    assert errs[True] < 0.5 * errs[False]


def test_gather_add_centred():
    """pcb_gather_add_bf16 with a centre: rows and statistics of u[idx] + v - centre."""
    from pointcloud_bridge_amd import _lib
    from pointcloud_bridge_amd.ops import _launch
    g = torch.Generator().manual_seed(4)
    B, N, S, ns, C = 2, 96, 40, 8, 72
    R = B * S * ns
    u = (torch.randn(B * N, C, generator=g) + 4).cuda()
    v = torch.randn(B * S, C, generator=g).cuda()
    idx = torch.randint(0, N, (B, S, ns), generator=g).cuda()
    centre = (torch.randn(C, generator=g) * 0.1 + 4).cuda()
    npart = _lib.load().pcb_gather_add_partials(R, C)
    y = torch.empty(R, C, dtype=torch.bfloat16, device="cuda")
    slabs = torch.empty(npart, 2, C, device="cuda")
    _launch("pcb_gather_add_bf16", 0, u.data_ptr(), v.data_ptr(), idx.data_ptr(), B, N, S, ns, C, 0, 0, 0, 3, y.data_ptr(),
            slabs.data_ptr(), npart, centre.data_ptr())
    src = (idx + torch.arange(B, device="cuda").view(B, 1, 1) * N).reshape(-1)
    ref = (u[src] + v[torch.arange(B * S, device="cuda").repeat_interleave(ns)]) - centre
    assert torch.equal(y, ref.to(torch.bfloat16))
    torch.testing.assert_close(slabs[:, 0].sum(0), y.float().sum(0), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(slabs[:, 1].sum(0), (y.float() ** 2).sum(0), rtol=1e-4, atol=1e-2)
