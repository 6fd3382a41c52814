"""GPU: the bf16 row kernels (csrc/rowbn.hip) and the bf16 execution mode of the networks.

Each kernel is compared with a plain PyTorch fp32 evaluation of the same operator on the same
(bf16-rounded) inputs.  Tolerances are bf16's: outputs are rounded to 8 significant bits once,
statistics and parameter gradients are fp32.  The 1e-4 logit bar belongs to the fp32 mode
(tests/test_gpu_modules.py); here the bf16 networks are checked against the fp32 networks and on a
short training run (mIoU parity on a learnable synthetic task).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rm():
    assert torch.cuda.is_available()
    from pointcloud_bridge_amd import rowmlp
    yield rowmlp
    rowmlp.set_precision("fp32")


def _act(u, act):
    return F.relu(u) if act == 1 else (F.leaky_relu(u, 0.2) if act == 2 else u)


def _ste_bf16(y):
    """Round to bf16 in the forward, identity in the backward (the kernels store y as bf16)."""
    return y + (y.to(torch.bfloat16).float() - y).detach()


def _ste_centred(z):
    """What a fresh training-mode layer stores (rowmlp._centre_args, pcb_gemm_nt_stats_bf16): the first call learns the
    batch mean c of the bf16-rounded product (probe) and stores y = bf16(z - c); everything downstream works on y, and
    BatchNorm does not see the constant.  Forward: that rounding, expressed back in the uncentred frame; backward: identity."""
    c = z.detach().to(torch.bfloat16).float().mean(0)
    return z + (((z.detach() - c).to(torch.bfloat16).float() + c) - z).detach()


def _first_max_pool(h, pool):
    """max over each `pool` consecutive rows with the gradient routed to the FIRST maximal row (what
    the kernels do; torch.max leaves the choice among equal values open, and equal bf16 values are
    common)."""
    h3 = h.view(-1, pool, h.shape[1])
    eq = h3 == h3.max(dim=1, keepdim=True)[0]
    first = eq & (eq.cumsum(dim=1) == 1)
    return (h3 * first).sum(dim=1)


def _close(a, b, tol):
    # bf16 operands: mean relative error at `tol`; 99.9 % of the elements within 4*tol of the largest.
    # (The remaining 0.1 %: a max-pool winner may change when two candidates differ by less than the
    # rounding of the folded scale/shift, which moves that group's gradient to another row.)
    d = (a.float() - b).abs()
    flat = d.flatten()
    q = float(flat.kthvalue(max(1, int(flat.numel() * 0.999)))[0]) if flat.numel() > 1000 else float(d.max())
    ok = (float(d.mean()) <= tol * max(float(b.abs().mean()), 1e-6)
          and q <= 4 * tol * max(float(b.abs().max()), 1e-6))
    if not ok:
        print(f"_close failed: mean |d| {float(d.mean()):.4g} vs mean |b| {float(b.abs().mean()):.4g}; "
              f"max |d| {float(d.max()):.4g} vs max |b| {float(b.abs().max()):.4g}; shape {tuple(b.shape)}")
    return ok


@pytest.mark.parametrize("R,K,C,act,pool", [(4096, 8, 64, 1, 0), (2048, 264, 128, 1, 16), (1536, 128, 64, 2, 8),
                                            (999 * 4, 72, 256, 0, 0), (640, 520, 512, 1, 32), (77 * 3, 16, 24, 1, 3),
                                            (1024, 264, 256, 1, 8), (1000, 264, 256, 2, 0)])
def test_linear_bn_act_forward_backward_vs_torch_fp32(rm, R, K, C, act, pool):
    torch.manual_seed(R + C)
    dev = "cuda"
    x = torch.randn(R, K, device=dev).to(torch.bfloat16).requires_grad_(True)
    conv = nn.Conv1d(K, C, 1).to(dev)
    bn = nn.BatchNorm1d(C).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    bn_ref = nn.BatchNorm1d(C).to(dev).train()
    bn_ref.load_state_dict(bn.state_dict())

    with rm.precision("bf16"):
        out = rm.conv_bn_act(conv, bn, x, act, pool)
    assert out.dtype == torch.bfloat16 and out.shape == (R // pool if pool else R, C)

    # reference: fp32 math on the same bf16-rounded operands; y rounded to bf16 as the kernels store it
    # (the conv bias cancels inside a train-mode BatchNorm, so it is added after the rounding)
    xr = x.detach().float().requires_grad_(True)
    w = conv.weight.detach().view(C, K).to(torch.bfloat16).float().requires_grad_(True)
    y = _ste_centred(xr @ w.t()) + conv.bias.detach()
    ref = _act(bn_ref(y), act)
    if pool:
        ref = _first_max_pool(ref, pool)
    scale = float(ref.detach().abs().max())
    assert float((out.float() - ref).abs().max()) < 2e-2 * scale
    torch.testing.assert_close(bn.running_mean, bn_ref.running_mean, rtol=1e-2, atol=2e-3)
    torch.testing.assert_close(bn.running_var, bn_ref.running_var, rtol=2e-2, atol=2e-3)
    assert int(bn.num_batches_tracked) == 1

    g = torch.randn_like(ref)
    (out.float() * g).sum().backward()
    (ref * g).sum().backward()
    assert _close(x.grad, xr.grad, 2e-2)
    assert _close(conv.weight.grad.view(C, K), w.grad, 2e-2)
    assert _close(bn.weight.grad, bn_ref.weight.grad, 2e-2)
    assert _close(bn.bias.grad, bn_ref.bias.grad, 2e-2)
    assert float(conv.bias.grad.abs().max()) == 0.0  # exactly zero under batch statistics


@pytest.mark.parametrize("R,K,widths,act,pool,perm", [(2048, 8, [64, 64, 128], 1, 16, 0), (1024, 264, [128, 128, 256], 1, 32, 0),
                                                      (3000, 72, [256, 128], 1, 0, 0), (1280, 128, [64], 2, 20, 0),
                                                      (1536, 24, [32, 32, 64], 1, 8, 16), (1100, 264, [256, 256], 1, 0, 0),
                                                      # dy written out once (C > 256 with > 256 inputs): top layer / inner layer / single layer
                                                      (900, 384, [512, 320], 1, 0, 0), (700, 320, [384, 64], 1, 0, 0), (600, 264, [384], 2, 0, 0)])
def test_fused_stack_forward_backward_vs_torch_fp32(rm, R, K, widths, act, pool, perm):
    """Multi-layer stack: BatchNorm+activation applied on operand load, statistics from the GEMM
    epilogue, dy recomputed on load in both backward GEMMs -- against a plain fp32 torch stack."""
    torch.manual_seed(R + K)
    dev = "cuda"
    x = torch.randn(R, K, device=dev).to(torch.bfloat16).requires_grad_(True)
    kin = K if not perm else perm + 3     # real input width when the rows carry padding
    dims = [kin] + widths
    convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip(dims[:-1], dims[1:])).to(dev)
    bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).to(dev).train()
    refs = nn.ModuleList(nn.BatchNorm1d(b) for b in widths).to(dev).train()
    with torch.no_grad():
        for bn, rf in zip(bns, refs):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            rf.load_state_dict(bn.state_dict())
        if perm:
            x.data[:, kin:] = 0          # padding columns as pcb_group_rows_bf16 writes them
    with rm.precision("bf16"):
        out = rm.mlp_rows(convs, bns, x, act, pool, perm)

    xr = x.detach().float().requires_grad_(True)
    h = xr[:, :kin]
    if perm:  # rows are [features | xyz]; the reference weight expects [xyz | features]
        h = torch.cat([h[:, perm:perm + 3], h[:, :perm]], dim=1)
    ws = []
    for li, (conv, rf) in enumerate(zip(convs, refs)):
        w = conv.weight.detach().view(conv.out_channels, -1).to(torch.bfloat16).float().requires_grad_(True)
        ws.append(w)
        h = _act(rf(_ste_centred(h @ w.t()) + conv.bias.detach()), act)
        if li < len(convs) - 1:
            h = _ste_bf16(h)  # the next GEMM consumes the activation as bf16
    ref = _first_max_pool(h, pool) if pool else h
    assert out.shape == ref.shape
    assert float((out.float() - ref).abs().max()) < 2e-2 * float(ref.detach().abs().max())
    g = torch.randn_like(ref)
    (out.float() * g).sum().backward()
    (ref * g).sum().backward()
    assert _close(x.grad[:, :kin], xr.grad[:, :kin], 2e-2)
    for conv, w, bn, rf in zip(convs, ws, bns, refs):
        assert _close(conv.weight.grad.view_as(w), w.grad, 2e-2)
        assert _close(bn.weight.grad, rf.weight.grad, 2e-2)
        assert _close(bn.bias.grad, rf.bias.grad, 2e-2)
        torch.testing.assert_close(bn.running_var, rf.running_var, rtol=3e-2, atol=3e-3)


def test_eval_mode_uses_running_stats_and_bias(rm):
    torch.manual_seed(0)
    dev = "cuda"
    R, K, C = 1024, 32, 64
    x = torch.randn(R, K, device=dev)
    conv = nn.Conv1d(K, C, 1).to(dev)
    bn = nn.BatchNorm1d(C).to(dev)
    with torch.no_grad():
        bn.running_mean.uniform_(-1, 1)
        bn.running_var.uniform_(0.5, 2)
    bn.eval()
    with rm.precision("bf16"):
        out = rm.conv_bn_act(conv, bn, x, 1)
    ref = F.relu(bn(F.linear(x.to(torch.bfloat16).float(), conv.weight.view(C, K).to(torch.bfloat16).float(), conv.bias)))
    assert float((out.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())


def test_group_rows_bf16_layout_and_backward(rm):
    from pointcloud_bridge_amd import ops
    torch.manual_seed(1)
    B, N, S, ns, C = 2, 300, 40, 8, 16
    xyz = torch.rand(B, N, 3, device="cuda")
    new_xyz = xyz[:, :S].contiguous()
    feat = torch.randn(B, N, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    idx = torch.randint(0, N, (B, S, ns), device="cuda")
    with rm.precision("bf16"):
        rows, perm = rm.group_rows(xyz, new_xyz, feat, idx)
    assert perm == C and rows.shape == (B * S * ns, 24) and rows.dtype == torch.bfloat16
    ref = ops.group_points(xyz, new_xyz, feat.detach().float(), idx).view(-1, 3 + C)
    assert torch.equal(rows[:, :C].float(), ref[:, 3:])                       # features first, exact copy
    assert torch.equal(rows[:, C:C + 3], ref[:, :3].to(torch.bfloat16))       # then centred coordinates
    assert float(rows[:, C + 3:].abs().max()) == 0.0                          # zero padding
    g = torch.randn(rows.shape, device="cuda").to(torch.bfloat16)
    rows.backward(g)
    want = torch.zeros(B, N, C, device="cuda")
    want.view(-1, C).index_add_(0, (idx + torch.arange(B, device="cuda").view(B, 1, 1) * N).view(-1), g[:, :C].float())
    torch.testing.assert_close(feat.grad.float(), want, rtol=2e-2, atol=2e-2)


def _build(cls, *a, **kw):
    torch.manual_seed(42)
    return cls(*a, **kw).cuda()


@pytest.mark.parametrize("name", ["pn2_ssg", "pn2_msg", "dgcnn"])
def test_bf16_networks_track_fp32_networks(rm, name):
    import bench
    torch.manual_seed(3)
    B, N = 2, (1024 if name == "dgcnn" else 2048)
    xyz, colors, labels = bench.synthetic_batch(B, N, 5, "cuda")
    res = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(42)
        model, cdim = bench.build_model(name)
        model = model.cuda().train()
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.eval()
        with rm.precision(mode):
            torch.manual_seed(9)
            logits = model(xyz, colors)
            loss = bench.loss_fn(logits, labels, cdim)
            loss.backward()
        gn = torch.stack([p.grad.norm() for p in model.parameters() if p.grad is not None])
        res[mode] = (logits.detach().float(), float(loss), gn)
    l32, l16 = res["fp32"][0], res["bf16"][0]
    assert l16.shape == l32.shape
    # bf16 activations through ~20 layers; single points can move (arg-max / neighbour flips), the
    # bulk must not
    print(name, "mean rel", float((l16 - l32).abs().mean() / l32.abs().mean()),
          "loss", res["bf16"][1], res["fp32"][1])
    assert float((l16 - l32).abs().mean() / l32.abs().mean()) < 0.35
    assert abs(res["bf16"][1] - res["fp32"][1]) < 0.02 * res["fp32"][1]
    big = res["fp32"][2] > 1e-3 * res["fp32"][2].max()
    # DGCNN rebuilds its kNN graphs from the features: bf16 features give other neighbours than fp32 ones, i.e. another
    # function (logits 30 % apart at this size), and the gradient norms of the EdgeConv blocks move by 10-40 % with
    # ANY change of the rounding (measured, tools/centre_debug3.py: 0.78-1.15 of the fp32 norms with uncentred rows,
    # 1.0-1.45 with centred ones on this batch; against the REFERENCE fixture the centred rows are the closer ones:
    # median 2.7e-2 against 1.4e-1, tools/bf16_parity.py).  The static-graph networks keep the tight bar.
    bar = 0.3 if name == "dgcnn" else 0.05
    assert float(((res["bf16"][2] - res["fp32"][2]).abs() / res["fp32"][2])[big].median()) < bar


def test_bf16_training_reaches_fp32_miou(rm):
    """Short trainings on a learnable synthetic task: mIoU (inference.py:814-855 definition) of the bf16 engine against
    the fp32 engine, in the REPRODUCIBLE mode (ops.set_deterministic: no float atomics anywhere in the step).

    Round 2 ran this once per mode with atomics and needed a 7-point bar, blaming the atomics' run-to-run drift (+-2.5
    points).  With the drift gone (a run repeated here gives the same mIoU to the last bit) the picture is: the
    DIFFERENCE between the two engines at a given seed and step is itself a chaotic quantity -- SGD on 8 scenes
    amplifies any rounding difference, the validation mIoU of either engine moves by +-2 points from one checkpoint to
    the next, and another fp32 summation order in ONE kernel of the library moves every sample -- while its mean is
    small and stable: tools/miou_det.py, 6 model seeds x 4 checkpoints (120 ... 960 steps): -0.4, -0.6, -1.0, -1.2 points
    per checkpoint, -0.8 +- 0.4 overall, single samples between -3.2 and +4.5.  The bf16 engine trails the fp32 engine by
    about a point on this task and does not drift away with training; resolving +-0.5 points would take
    ~100 trainings per engine.  Asserted over 18 samples: |mean| below 2 points (three standard errors), every sample within 12 (the samples'
    spread is 2.8 points: 8 points is a 3-sigma event that one set of 18 in fifteen contains -- seen after the dropout
    kernel changed the trajectories), both engines
    above 0.7 mIoU at the last checkpoint (chance 0.2), and a repeated run identical to the bit."""
    from pointcloud_bridge_amd import ops, train
    from pointcloud_bridge_amd.models.containers import PointNet2
    from pointcloud_bridge_amd.models.pointnet2_utils import FeaturePropagation
    enc = [(256, 0.2, 16, 6, [32, 32, 64]), (64, 0.4, 16, 67, [64, 64, 128]), (16, 0.8, 16, 131, [128, 128, 256])]
    data = train.synthetic_scenes(8, 1024, seed=0, device="cuda")
    val = train.synthetic_scenes(4, 1024, seed=1, device="cuda")
    checkpoints = (120, 240, 480)

    def run(mode, seed):
        torch.manual_seed(seed)
        model = PointNet2(5, encoder=enc)
        # decoder widths follow the encoder of this small variant
        model.fp3 = FeaturePropagation(256 + 128, [128, 128])
        model.fp2 = FeaturePropagation(128 + 64, [128, 64])
        model.fp1 = FeaturePropagation(64, [128, 128, 128])
        model = model.cuda()
        with rm.precision(mode):
            tr = train.Trainer(model, 5, lr=2e-3)
            torch.manual_seed(0)
            torch.cuda.manual_seed(0)
            hist = []
            for i in range(max(checkpoints)):
                tr.train_step(data)
                if i + 1 in checkpoints:
                    hist.append(tr.evaluate([val])["miou"])
        return hist

    old = ops.set_deterministic(True)
    try:
        diffs = []
        for seed in (42, 43, 44, 45, 46, 47):
            f32, b16 = run("fp32", seed), run("bf16", seed)
            print("seed", seed, "fp32", [round(v, 4) for v in f32], "bf16", [round(v, 4) for v in b16])
            assert min(f32[-1], b16[-1]) > 0.7    # both engines learn the task (chance: 0.2)
            diffs += [b - f for f, b in zip(f32, b16)]
        assert run("bf16", 47) == b16             # reproducible: the same trajectory, bit for bit
    finally:
        ops.set_deterministic(old)
    n = len(diffs)
    mean = sum(diffs) / n
    se = (sum((d - mean) ** 2 for d in diffs) / (n - 1)) ** 0.5 / n ** 0.5
    print("bf16 - fp32 mIoU over", n, "samples: mean", round(mean, 4), "standard error", round(se, 4), "min", round(min(diffs), 4),
          "max", round(max(diffs), 4))
    assert abs(mean) < 0.02
    assert max(abs(d) for d in diffs) < 0.12


@pytest.mark.parametrize("D1,C,k", [(3, 64, 4), (0, 32, 3), (16, 256, 4), (5, 8, 3)])
def test_interpolate_concat_bf16_forward_backward(rm, D1, C, k):
    """bf16 interpolate+concat buffer (gap layout) and its CSR backward against the fp32 operator."""
    from pointcloud_bridge_amd import ops
    torch.manual_seed(D1 + C)
    B, N, S = 2, 700, 90
    g = torch.Generator().manual_seed(1)
    xyz1 = torch.rand(B, N, 3, generator=g).cuda()
    xyz2 = torch.rand(B, S, 3, generator=g).cuda()
    feat = torch.randn(B, S, C, generator=g).cuda().to(torch.bfloat16).requires_grad_(True)
    skip = torch.randn(B * N, D1, generator=g).cuda().to(torch.bfloat16).requires_grad_(True) if D1 else None
    d2, idx = ops.three_nn(xyz1, xyz2, k)
    with rm.precision("bf16"):
        rows, perm = rm.interpolate_concat(skip, feat, d2, idx)
    dp = (D1 + 7) // 8 * 8
    assert rows.shape == (B * N, dp + C) and perm == (-D1 if D1 % 8 else 0)
    ref_feat = feat.detach().float().requires_grad_(True)
    ref = ops.three_interpolate(ref_feat, d2, idx).view(B * N, C)
    assert float((rows[:, dp:].float() - ref).abs().max()) < 1e-2 * float(ref.abs().max())
    if D1:
        assert torch.equal(rows[:, :D1], skip.detach())
        assert float(rows[:, D1:dp].abs().max()) == 0.0 if dp > D1 else True
    gr = torch.randn(rows.shape, device="cuda").to(torch.bfloat16)
    rows.backward(gr)
    (ref * gr[:, dp:].float()).sum().backward()
    d = (feat.grad.float() - ref_feat.grad).abs()
    assert float(d.mean()) < 1e-2 * float(ref_feat.grad.abs().mean())
    if D1:
        assert torch.equal(skip.grad, gr[:, :D1])


def _unpad_weight_grad(dwp, k, perm, q=8):
    """Inverse of rowmlp.padded_weight_from's column layout for a weight gradient [Cout, kp] -> [Cout, k]."""
    if perm > 0:
        return torch.cat([dwp[:, perm:perm + 3], dwp[:, :perm]], dim=1)
    if perm < 0:
        d, dp = -perm, (-perm + q - 1) // q * q
        return torch.cat([dwp[:, :d], dwp[:, dp:dp + k - d]], dim=1)
    return dwp[:, :k]


@pytest.mark.parametrize("perm,k,kp", [(0, 6, 8), (0, 64, 64), (3, 6, 8), (128, 131, 136), (-3, 259, 264), (-5, 13, 16)])
def test_prep_weights_and_wgrad_layout_match_host_reference(rm, perm, k, kp):
    """pcb_prep_weights_bf16 == padded_weight_from (+ transpose); pcb_gemm_tn_bf16's out_cols/out_perm
    output == _unpad_weight_grad of the padded result (bitwise: same sums, other placement)."""
    import ctypes
    from pointcloud_bridge_amd import _lib
    from pointcloud_bridge_amd.ops import _launch
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(11)
    C, R = 64, 4096
    w = torch.randn(C, k, 1, 1, generator=g).to(dev)
    ref = rm.padded_weight_from(w, kp, perm)
    wp = torch.empty(C, kp, dtype=torch.bfloat16, device=dev)
    wt = torch.empty(kp, C, dtype=torch.bfloat16, device=dev)
    desc = (ctypes.c_longlong * 8)(w.data_ptr(), wp.data_ptr(), wt.data_ptr(), C, k, kp, perm, 0)
    _launch("pcb_prep_weights_bf16", 0, 1, desc)
    assert torch.equal(wp, ref) and torch.equal(wt, ref.t().contiguous())

    dz = torch.randn(R, C, generator=g).to(dev).to(torch.bfloat16)
    x = torch.randn(R, kp, generator=g).to(dev).to(torch.bfloat16)
    lib = _lib.load()
    ws = torch.empty(lib.pcb_gemm_tn_workspace(R, C, kp), dtype=torch.float32, device=dev)
    full = torch.empty(C, kp, dtype=torch.float32, device=dev)
    real = torch.full((C, k), float("nan"), dtype=torch.float32, device=dev)
    for out, cols, pm in ((full, 0, 0), (real, k, perm)):
        _launch("pcb_gemm_tn_bf16", 0, 0, dz.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, x.data_ptr(), 0, 0, 0, R, C, kp,
                ws.data_ptr(), out.data_ptr(), cols, pm)
    assert torch.equal(real, _unpad_weight_grad(full, k, perm).contiguous())
    assert _close(full, dz.float().t() @ x.float(), 2e-3)


def test_stack_call_equals_layer_by_layer_calls(rm):
    """pcb_mlp_stack_forward/backward enqueue exactly the per-layer entry points: a 3-layer stack run
    as ONE stack equals the same layers run as three 1-layer stacks chained through autograd, up to
    the one extra bf16 rounding of the materialised activations between them."""
    rm.set_precision("bf16")
    dev = torch.device("cuda")
    torch.manual_seed(5)
    widths, K, R, pool = [64, 64, 128], 8, 4096, 16
    convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip([K] + widths[:-1], widths)).to(dev)
    bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).to(dev)
    x = torch.randn(R, K, device=dev).to(torch.bfloat16).requires_grad_(True)

    def run(fused):
        for m in list(convs) + list(bns):
            m.zero_grad(set_to_none=True)
        for b in bns:
            b.reset_running_stats()
        rm.reset_centres(bns)
        x.grad = None
        if fused:
            out = rm.mlp_rows(convs, bns, x, pool=pool)
        else:
            h = x
            for i, (c, b) in enumerate(zip(convs, bns)):
                h = rm.conv_bn_act(c, b, h, pool=pool if i == len(convs) - 1 else 0)
            out = h
        out.float().square().sum().backward()
        assert all(int(b.num_batches_tracked) == 1 for b in bns)  # bumped by the finalize kernel, once
        return (out.detach().float(), x.grad.float(), [c.weight.grad.clone() for c in convs],
                [b.weight.grad.clone() for b in bns], [b.running_var.clone() for b in bns])

    a, b = run(True), run(False)
    assert _close(a[0], b[0], 1e-2) and _close(a[1], b[1], 3e-2)
    for u, v in zip(a[2] + a[3], b[2] + b[3]):
        assert _close(u, v, 3e-2)
    for u, v in zip(a[4], b[4]):
        assert torch.allclose(u, v, rtol=2e-2, atol=1e-4)
    rm.set_precision("fp32")


def test_fusion_on_coarse_rows_matches_full_resolution(rm):
    """MultiScaleFeatureFusion in bf16 mode runs Conv+BN+ReLU on the coarse rows and repeats the
    output; that must equal the reference order (F.interpolate to N points first, models/model.py:164)
    computed by the same kernels (coarse_rows = False), including the running statistics, and stay
    within bf16 distance of the fp32 mode."""
    from pointcloud_bridge_amd.models.containers import MultiScaleFeatureFusion
    dev = torch.device("cuda")
    torch.manual_seed(3)
    B, n = 2, 512
    levels = [(32, 64), (32, 128), (16, n)]
    fus = MultiScaleFeatureFusion([c for c, _ in levels], 16).to(dev).train()
    feats = [torch.randn(B, c, s, device=dev).to(torch.bfloat16).float() for c, s in levels]

    def run(mode, coarse):
        rm.set_precision(mode)
        fus.coarse_rows = coarse
        for conv in fus.convs:
            conv[1].reset_running_stats()
        rm.reset_centres(fus)
        fus.zero_grad(set_to_none=True)
        xs = [f.clone().requires_grad_(True) for f in feats]
        out = fus(xs)
        (out.float() * torch.linspace(-1, 1, out.shape[-1], device=dev)).sum().backward()
        return (out.detach().float(), [x.grad for x in xs], [c[0].weight.grad.clone() for c in fus.convs],
                [c[1].running_var.clone() for c in fus.convs], [c[1].running_mean.clone() for c in fus.convs])

    try:
        fp32, full, got = run("fp32", True), run("bf16", False), run("bf16", True)
    finally:
        rm.set_precision("fp32")
        fus.coarse_rows = True
    assert got[0].shape == full[0].shape == (B, n, 48)
    # same kernels, same rounding: only the summation order of the statistics differs
    assert _close(got[0], full[0], 2e-3)
    for u, v in zip(got[1] + got[2], full[1] + full[2]):
        assert _close(u, v, 5e-3)
    for u, v in zip(got[3] + got[4], full[3] + full[4]):
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-6)   # the unbiased count is B*n in both
    assert _close(got[0], fp32[0], 1e-2)
    for u, v in zip(got[3] + got[4], fp32[3] + fp32[4]):
        assert torch.allclose(u, v, rtol=5e-3, atol=2e-4)


def _run_module(mod, args, grad_of, rm, gathered, mode="bf16"):
    rm.set_precision(mode)
    rm.set_gathered(gathered)
    mod.zero_grad(set_to_none=True)
    for m in mod.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            m.reset_running_stats()
    rm.reset_centres(mod)
    torch.manual_seed(1)  # FPS start indices: the same draw in every run
    leaf = grad_of.clone().requires_grad_(True)
    out = mod(*[leaf if a is grad_of else a for a in args])
    out = out[1] if isinstance(out, tuple) else out
    w = torch.linspace(-1.0, 1.0, out.numel(), device=out.device).view_as(out)
    (out.float() * w).sum().backward()
    res = {"out": out.detach().float(), "dx": leaf.grad.float()}
    res.update({"d" + n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None})
    res.update({n: b.clone().float() for n, b in mod.named_buffers() if "running" in n})
    return res


def _rel(a, b):
    return float((a - b).abs().mean() / b.abs().mean().clamp_min(1e-12))


@pytest.mark.parametrize("kind", ["sa", "msg", "edge"])
def test_gathered_first_layer_matches_grouped_rows(rm, kind):
    """The per-point evaluation of a grouped stack's first conv (csrc/gatherlin.hip) against the
    grouped-rows path of the same bf16 engine and against the fp32 mode: outputs, input/parameter
    gradients, running statistics.  With so few rows a ReLU decision that flips under bf16 rounding
    moves a parameter gradient by percents in EITHER bf16 path, so the bar is relative: the gathered
    path must sit as close to the fp32 result as the grouped path does (it keeps the first layer's
    operands in fp32, so usually closer), and both bf16 paths must agree on the forward values."""
    from pointcloud_bridge_amd.models import pointnet2_utils as pu
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    dev = torch.device("cuda")
    torch.manual_seed(9)
    B, N = 2, 512
    xyz = torch.rand(B, N, 3, device=dev) * 2 - 1
    try:
        if kind == "edge":
            net = DGCNN(5, k=8).to(dev).train()
            x = torch.randn(B, N, 64, device=dev).to(torch.bfloat16).float()
            mod = nn.Module()
            mod.block = net.conv2
            mod.forward = lambda t: net._edge_conv(net.conv2, t, 8)
            args, leaf = (x,), x
        else:
            feat = torch.randn(B, 64, N, device=dev).to(torch.bfloat16).float()
            if kind == "sa":
                mod = pu.SetAbstraction(64, 0.4, 16, 64 + 3, [32, 32, 64]).to(dev).train()
            else:
                mod = pu.MultiScaleSetAbstraction(64, [0.3, 0.6], [8, 16], 64 + 3, [32, 32, 64]).to(dev).train()
            args, leaf = (xyz, feat), feat
        gath = _run_module(mod, args, leaf, rm, True)
        grp = _run_module(mod, args, leaf, rm, False)
        ref = _run_module(mod, args, leaf, rm, False, "fp32")
    finally:
        rm.set_gathered(True)
        rm.set_precision("fp32")
    assert set(gath) == set(grp) == set(ref)
    assert _rel(gath["out"], grp["out"]) < 1e-2
    for n in ref:
        eg, er = _rel(gath[n], ref[n]), _rel(grp[n], ref[n])
        assert eg <= max(1.6 * er, 3e-2), (n, eg, er)


@pytest.mark.parametrize("pooled,with_v,with_wx,C", [(0, 1, 0, 64), (1, 1, 0, 64), (0, 0, 1, 64), (1, 0, 1, 64),
                                                      (0, 1, 1, 192), (1, 1, 0, 520), (0, 0, 1, 8)])
def test_gather_add_and_scatter_dy_entry_points(rm, pooled, with_v, with_wx, C):
    """pcb_gather_add_bf16 / pcb_scatter_dy_bf16 against a direct PyTorch evaluation of their
    definitions (include/pcb_hip.h): rows, BatchNorm statistics slabs, du / dv / dWx."""
    import ctypes
    from pointcloud_bridge_amd import _lib
    from pointcloud_bridge_amd.ops import _launch
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(21)
    B, N, S, ns = 2, 96, 40, 8
    R = B * S * ns
    u = torch.randn(B * N, C, generator=g).to(dev)
    v = torch.randn(B * S, C, generator=g).to(dev) if with_v else None
    wx = torch.randn(C, 3, generator=g).to(dev) if with_wx else None
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    ctr = torch.rand(B, S, 3, generator=g).to(dev)
    idx = torch.randint(0, N, (B, S, ns), generator=g).to(dev)
    idx[:, :, ns // 2:] = idx[:, :, :1]  # padded tail, as ball query produces it
    lib = _lib.load()
    npart = lib.pcb_gather_add_partials(R, C)
    y = torch.empty(R, C, dtype=torch.bfloat16, device=dev)
    slabs = torch.empty(npart, 2, C, device=dev)
    _launch("pcb_gather_add_bf16", 0, u.data_ptr(), 0 if v is None else v.data_ptr(), idx.data_ptr(), B, N, S, ns, C,
            xyz.data_ptr(), ctr.data_ptr(), 0 if wx is None else wx.data_ptr(), 3, y.data_ptr(), slabs.data_ptr(), npart, 0)
    src = (idx + torch.arange(B, device=dev).view(B, 1, 1) * N).reshape(-1)
    grp = torch.arange(B * S, device=dev).repeat_interleave(ns)
    ref = u[src]
    if with_v:
        ref = ref + v[grp]
    diff = xyz.reshape(-1, 3)[src] - ctr.reshape(-1, 3)[grp]
    if with_wx:
        ref = ref + diff @ wx.t()
    assert torch.allclose(y.float(), ref.to(torch.bfloat16).float(), rtol=0, atol=2 ** -7 * float(ref.abs().max()))
    yf = y.float()
    assert torch.allclose(slabs[:, 0].sum(0), yf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(slabs[:, 1].sum(0), (yf * yf).sum(0), rtol=1e-4, atol=1e-2)

    # backward: dy = scale*dz*act'(y*scale+shift) + p*y + q   (ReLU)
    scale, shift = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev) * 0.3
    p, q = torch.randn(C, generator=g).to(dev) * 0.1, torch.randn(C, generator=g).to(dev) * 0.1
    if pooled:
        dout = torch.randn(B * S, C, generator=g).to(dev)
        arg = torch.randint(0, ns, (B * S, C), generator=g).to(dev).to(torch.uint8)
        dzf = torch.zeros(B * S, ns, C, device=dev)
        dzf.scatter_(1, arg.long().unsqueeze(1), dout.unsqueeze(1))
        dzf = dzf.view(R, C)
        dz = None
    else:
        dz = torch.randn(R, C, generator=g).to(dev).to(torch.bfloat16)
        dzf, dout, arg = dz.float(), None, None
    dy = scale * dzf * ((yf * scale + shift) > 0).float() + p * yf + q
    du = torch.zeros(B * N, C, device=dev)
    dv = torch.empty(B * S, C, device=dev)
    dwx = torch.zeros(33, C, 3, device=dev) if with_wx else None
    _launch("pcb_scatter_dy_bf16", 0, pooled, 0 if dz is None else dz.data_ptr(), y.data_ptr(), scale.data_ptr(),
            shift.data_ptr(), p.data_ptr(), q.data_ptr(), 0 if dout is None else dout.data_ptr(),
            0 if arg is None else arg.data_ptr(), 1, idx.data_ptr(), B, N, S, ns, C, xyz.data_ptr(), ctr.data_ptr(),
            du.data_ptr(), dv.data_ptr(), 0 if dwx is None else dwx.data_ptr(), 0)
    du_ref = torch.zeros(B * N, C, device=dev).index_add_(0, src, dy)
    dv_ref = dy.view(B * S, ns, C).sum(1)
    assert torch.allclose(du, du_ref, rtol=1e-4, atol=1e-4)
    assert torch.allclose(dv, dv_ref, rtol=1e-4, atol=1e-4)
    if with_wx:
        assert torch.allclose(dwx[0], dy.t() @ diff, rtol=1e-4, atol=1e-3)


def test_gate_rows_matches_torch(rm):
    """pcb_gate_bf16 / pcb_gate_bwd_bf16 == x * sigmoid(a) and its autograd gradients (bf16 rounding)."""
    dev = torch.device("cuda")
    torch.manual_seed(2)
    x = torch.randn(777, 264, device=dev).to(torch.bfloat16)
    a = (torch.randn(777, 264, device=dev) * 3).to(torch.bfloat16)
    g = torch.randn(777, 264, device=dev).to(torch.bfloat16)
    with rm.precision("bf16"):
        x1, a1 = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
        out = rm.gate_rows(x1, a1)
        out.backward(g)
    x2, a2 = x.float().requires_grad_(True), a.float().requires_grad_(True)
    ref = x2 * torch.sigmoid(a2)
    ref.backward(g.float())
    assert out.dtype == torch.bfloat16
    assert torch.allclose(out.float(), ref, rtol=2 ** -7, atol=1e-3)
    assert torch.allclose(x1.grad.float(), x2.grad, rtol=2 ** -7, atol=1e-3)
    assert torch.allclose(a1.grad.float(), a2.grad, rtol=2 ** -7, atol=1e-3)


def test_eval_operand_cache_is_transparent_and_invalidated_by_updates():
    """No-grad eval-mode calls keep the prepared bf16 operands / BatchNorm constants of a stack.  The
    second call (cache hit) must equal the first bit for bit, and a parameter or running-statistics
    update must be picked up (version counters), as must a fresh module at the same address."""
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    with rowmlp.precision("bf16"):
        torch.manual_seed(1)
        g = torch.Generator().manual_seed(3)
        v = torch.randn(2, 2048, 3, generator=g)
        xyz = (v / v.norm(dim=-1, keepdim=True) * torch.rand(2, 2048, 1, generator=g) ** (1 / 3)).cuda()
        colors = torch.rand(2, 2048, 3, generator=g).cuda()
        model = PointNet2MSG(5).cuda().eval()

        def run():
            torch.manual_seed(9)  # FPS start indices
            with torch.no_grad():
                return model(xyz, colors).float()

        a = run()
        b = run()
        assert torch.equal(a, b)
        with torch.no_grad():   # what an optimizer step / a train-mode forward does: in-place updates
            model.sa1.conv_blocks[0][0].weight.mul_(1.5)
            model.fp1.mlp_bns[0].running_mean.add_(0.25)
            model.final_fusion[4].bias.add_(1.0)
        c = run()
        assert float((c - a).abs().max()) > 1e-2
        rowmlp._eval_operands.clear()
        d = run()               # the same weights prepared afresh
        assert torch.equal(c, d)


def test_training_with_side_stream_prefetch_equals_training_without():
    """Two forward/backward passes with the next batch's coordinate-only work (FPS pyramid, ball queries,
    decoder k-NN, inverted index of the interpolation) prefetched on the side stream into the
    module-owned double buffers: the logits must equal the plain sequence bit for bit, the gradients up
    to the summation order of the scatter atomics (the plain sequence repeated differs from itself by
    the same amount)."""
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    with rowmlp.precision("bf16"):
        g = torch.Generator().manual_seed(4)
        batches = []
        for _ in range(2):
            v = torch.randn(2, 2048, 3, generator=g)
            xyz = (v / v.norm(dim=-1, keepdim=True) * torch.rand(2, 2048, 1, generator=g) ** (1 / 3)).cuda()
            batches.append((xyz, torch.rand(2, 2048, 3, generator=g).cuda(), torch.randint(0, 5, (2, 2048), generator=g).cuda()))
        torch.manual_seed(2)
        model = PointNet2MSG(5).cuda().train()
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.eval()
        state = {k: v.clone() for k, v in model.state_dict().items()}

        def run(prefetch):
            model.load_state_dict(state)   # running statistics back to the start
            rowmlp.reset_centres(model)    # ... and the engine's own per-layer state (centred bf16 rows)
            torch.manual_seed(17)          # CPU generator: FPS start indices
            logits, grads = [], []
            for i, (xyz, colors, labels) in enumerate(batches):
                model.zero_grad(set_to_none=True)
                out = model(xyz, colors).float()
                if prefetch and i + 1 < len(batches):
                    model.prefetch(batches[i + 1][0])
                F.cross_entropy(out, labels).backward()
                logits.append(out.detach().clone())
                grads.append(torch.cat([p.grad.reshape(-1).float() for p in model.parameters() if p.grad is not None]))
            return logits, grads

        la, ga = run(False)
        lr, gr = run(False)
        lb, gb = run(True)
        for a, b in zip(la, lb):
            assert torch.equal(a, b)
        for a, r, b in zip(ga, gr, gb):
            noise = float((a - r).norm() / a.norm())
            assert float((a - b).norm() / a.norm()) <= 3 * noise + 1e-4, noise
