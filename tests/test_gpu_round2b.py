"""GPU: kernels added in the second half of round 2.

  * the one-pass cross entropy (csrc/loss.hip) against F.cross_entropy -- the call the reference's trainers make
    (train_MulSca_PN2.py:161 on [B,C,N], train_DGCNN.py:177-197 on [B*N,C]) -- loss and gradient, both layouts,
    ignored labels, a row stride wider than C
  * the BatchNorm-backward sums of a stack's top layer in slab mode (one slab per workgroup, no atomics) against
    the single-slab mode and an fp64 evaluation
  * the slab sums of a weight gradient with few slabs of a large matrix (the one-element-per-lane path)
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("layout", ["bcn", "bnc", "rows"])
@pytest.mark.parametrize("C", [5, 13])
def test_cross_entropy_matches_torch(layout, C):
    from pointcloud_bridge_amd.losses import cross_entropy
    torch.manual_seed(3)
    B, N = 3, 1500
    rows = (torch.randn(B * N, C, device="cuda") * 3).requires_grad_(True)
    ref_rows = rows.detach().clone().requires_grad_(True)
    labels = torch.randint(0, C, (B, N), device="cuda")
    labels[0, :17] = -100  # ignored points
    if layout == "bcn":   # what the PointNet++ containers return: a transposed view of the rows
        loss = cross_entropy(rows.view(B, N, C).transpose(1, 2), labels)
        ref = F.cross_entropy(ref_rows.view(B, N, C).transpose(1, 2), labels)
    elif layout == "bnc":  # DGCNN
        loss = cross_entropy(rows.view(B, N, C), labels, channels_last=True)
        ref = F.cross_entropy(ref_rows.view(B, N, C).reshape(-1, C), labels.reshape(-1))
    else:
        loss = cross_entropy(rows, labels.reshape(-1), channels_last=True)
        ref = F.cross_entropy(ref_rows, labels.reshape(-1))
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    (loss * 1.7).backward()
    (ref * 1.7).backward()
    assert torch.allclose(rows.grad, ref_rows.grad, rtol=1e-5, atol=1e-9)
    assert rows.grad[:17].abs().max() == 0  # ignored points get no gradient


def test_cross_entropy_strided_rows_and_fallback():
    """Logits that are a column slice of wider rows are read in place; CPU logits take the reference's own call."""
    from pointcloud_bridge_amd.losses import cross_entropy
    torch.manual_seed(4)
    wide = torch.randn(4096, 8, device="cuda")
    labels = torch.randint(0, 5, (4096,), device="cuda")
    got = cross_entropy(wide[:, :5], labels, channels_last=True)
    ref = F.cross_entropy(wide[:, :5].contiguous(), labels)
    assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref))
    cpu = cross_entropy(wide[:, :5].cpu(), labels.cpu(), channels_last=True)
    assert abs(float(cpu) - float(ref)) <= 1e-5 * abs(float(ref))


def test_cross_entropy_labels_elsewhere_and_out_of_range():
    """labels that do not live on the logits' GPU never reach the kernel (their raw address would be read by
    the GPU): the call goes to F.cross_entropy and fails the way the reference's call does -- a RuntimeError, no fault.
    Labels outside [0, C): counted as ignored by the kernel (documented); check_labels=True raises IndexError."""
    from pointcloud_bridge_amd.losses import cross_entropy
    torch.manual_seed(6)
    rows = torch.randn(2048, 5, device="cuda", requires_grad=True)
    labels = torch.randint(0, 5, (2048,), device="cuda")
    with pytest.raises(RuntimeError):
        cross_entropy(rows, labels.cpu(), channels_last=True)
    ok = cross_entropy(rows, labels, channels_last=True, check_labels=True)
    assert abs(float(ok) - float(F.cross_entropy(rows, labels))) <= 2e-6 * abs(float(ok))
    bad = labels.clone()
    bad[:7] = 9
    with pytest.raises(IndexError):
        cross_entropy(rows, bad, channels_last=True, check_labels=True)
    # unchecked: the 7 points are skipped exactly like ignore_index points
    skipped = labels.clone()
    skipped[:7] = -100
    assert float(cross_entropy(rows, bad, channels_last=True)) == float(cross_entropy(rows, skipped, channels_last=True))


@pytest.mark.parametrize("sfx,dtype", [("bf16", torch.bfloat16), ("f32", torch.float32)])
@pytest.mark.parametrize("R,C", [(50000, 128), (8192, 1536), (333, 64)])
def test_bwd_reduce_slabs(sfx, dtype, R, C):
    from pointcloud_bridge_amd import _lib
    if dtype == torch.float32 and C > 1024:
        C = 1024  # one lane per 16-byte vector of a row: 256 vectors = 1024 fp32 / 2048 bf16 columns at most
    lib = _lib.load()
    torch.manual_seed(5)
    dz = torch.randn(R, C, device="cuda").to(dtype)
    y = torch.randn(R, C, device="cuda").to(dtype)
    scale, shift = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3
    mean, invstd = torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5
    fn = getattr(lib, "pcb_bn_act_bwd_reduce_" + sfx)
    du = dz.double() * ((y.double() * scale.double() + shift.double()) > 0)
    want = torch.stack([du.sum(0), (du * (y.double() - mean.double()) * invstd.double()).sum(0)])
    tol = 2e-3 * float(want.abs().max()) if dtype == torch.float32 else 2e-3 * float(want.abs().max())
    for nparts in (1, 7, 768):
        sums = torch.zeros(nparts, 2, C, device="cuda") if nparts == 1 else torch.full((nparts, 2, C), 7.0, device="cuda")
        rc = fn(dz.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                R, C, 1, sums.data_ptr(), nparts, _stream())
        assert rc == 0
        got = sums.double().sum(0)
        assert float((got - want).abs().max()) <= tol, (nparts, float((got - want).abs().max()), tol)
    assert fn(dz.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
              R, C, 1, sums.data_ptr(), 769, _stream()) < 0  # more slabs than the library ever writes


@pytest.mark.parametrize("R,M,N", [(8192, 1536, 1024), (262144, 64, 64), (131072, 256, 264)])
def test_weight_gradient_slab_sums(R, M, N):
    """dW = dy^T x through pcb_gemm_tn_bf16: few slabs of a large matrix (first shape), hundreds of slabs of a small
    one (second) and many slabs of a large one (third: 16-byte loads) take different paths of the slab-sum kernel; all
    against an fp64 product."""
    from pointcloud_bridge_amd import _lib
    lib = _lib.load()
    torch.manual_seed(6)
    dy = (torch.randn(R, M, device="cuda") * 0.1).to(torch.bfloat16)
    x = (torch.randn(R, N, device="cuda") * 0.1).to(torch.bfloat16)
    ws = torch.empty(lib.pcb_gemm_tn_workspace(R, M, N), dtype=torch.float32, device="cuda")
    dW = torch.empty(M, N, dtype=torch.float32, device="cuda")
    rc = lib.pcb_gemm_tn_bf16(0, dy.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, x.data_ptr(), 0, 0, 0, R, M, N,
                              ws.data_ptr(), dW.data_ptr(), N, 0, _stream())
    assert rc == 0
    want = dy.double().t() @ x.double()
    assert float((dW.double() - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-4


@pytest.mark.parametrize("sfx,dtype,kp", [("bf16", torch.bfloat16, 8), ("f32", torch.float32, 4)])
def test_pad_rows(sfx, dtype, kp):
    """Raw fp32 coordinate / colour columns as a padded operand of the row type: equal to cast + F.pad."""
    from pointcloud_bridge_amd import rowmlp
    x = torch.randn(5000, 6, device="cuda")[:, :3]  # rows 6 floats apart
    with rowmlp.precision("bf16" if sfx == "bf16" else "fp32"):
        got = rowmlp._rows(x, kp, rowmlp.mode())
    want = F.pad(x.to(dtype), (0, kp - 3))
    assert got.dtype == dtype and torch.equal(got, want)


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_bench_on_a_one_rank_rccl_group(tmp_path, mode):
    """The gradient exchange on the REAL backend (`nccl` = RCCL) with one rank -- all this box can host:
    torch.distributed.run --nproc-per-node 1, PCB_DIST_SINGLE=1 makes bench.py join the group and issue every
    collective of the multi-GPU run (parameter broadcast, bucketed asynchronous all-reduce during the backward pass
    or the flat one behind the captured step, the barriers and max-reductions of the timing, the mode vote).  The
    losses equal those of the plain single-process run: a sum over one rank changes nothing."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PCB_DIST_BACKEND", None)
    common = ["--gpus", "1", "--steps", "3", "--warmup", "0" if mode == "eager" else "2", "--npoints", "2048", "--batch", "4",
              "--no-cpu-baseline", "--no-extras", "--no-dropout", "--exec", mode]
    if mode == "eager":
        common += ["--precision", "fp32"]  # two runs of the same fp32 command agree to ~2e-6 (tools/grad_noise.py); bf16 rows: 8e-3
    dumps = (str(tmp_path / "plain.pt"), str(tmp_path / "rccl.pt"))
    extra = (lambda i: ["--dump", dumps[i]]) if mode == "eager" else (lambda i: [])  # (a dump keeps the step eager)
    plain = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + common + extra(0), env=env,
                           capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stdout[-2000:] + plain.stderr[-4000:]
    from tests.helpers import free_port
    port = str(free_port())
    rccl = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(repo, "bench.py")] + common
                          + extra(1), env=dict(env, PCB_DIST_SINGLE="1"), capture_output=True, text=True, timeout=600)
    assert rccl.returncode == 0, rccl.stdout[-2000:] + rccl.stderr[-4000:]
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    b = json.loads(rccl.stdout.strip().splitlines()[-1])
    assert b["config"]["exec"]["mode"] == mode
    # (bf16 rows, scatter-adds by fp32 atomics: two runs of the SAME command differ by ~5e-4 after five Adam steps)
    assert abs(a["config"]["loss"] - b["config"]["loss"]) <= 1e-2 * abs(a["config"]["loss"]), (a["config"]["loss"], b["config"]["loss"])
    if mode == "eager":
        da, db = torch.load(dumps[0], weights_only=True), torch.load(dumps[1], weights_only=True)
        assert abs(da["losses"][0] - db["losses"][0]) <= 1e-6 * abs(da["losses"][0])  # the first forward pass: no update yet
        ga, gb = da["first_grad"], db["first_grad"]
        assert float((ga - gb).norm() / ga.norm()) < 1e-4  # the all-reduced gradient IS the local one


def _msg_pair():
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    torch.manual_seed(42)
    a = PointNet2MSG(5).cuda().train()
    torch.manual_seed(42)
    b = PointNet2MSG(5).cuda().train()
    for m in list(a.modules()) + list(b.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return a, b


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_prepare_step_changes_nothing_but_the_launch_count(precision):
    """rowmlp.prepare_step(): the operands of every stack from ONE launch after the optimiser step.  Two identical
    networks train three Adam steps on the same batches, one with prepare_step() after every update, one without;
    their eval logits then agree (fp32 rows: to rounding of the scatter-add order; bf16: to the run-to-run noise of
    that mode), the stacks of the first really skipped their own preparation, and an in-place edit of a weight
    after prepare_step() (version counter moves) is honoured by the next forward pass."""
    from pointcloud_bridge_amd import rowmlp
    a, b = _msg_pair()
    rowmlp.set_precision(precision)
    try:
        oa, ob = torch.optim.Adam(a.parameters(), lr=1e-3), torch.optim.Adam(b.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(1)
        B, N = 2, 2048
        for step in range(3):
            v = torch.randn(B, N, 3, generator=g)
            xyz = (v / v.norm(dim=-1, keepdim=True) * torch.rand(B, N, 1, generator=g) ** (1 / 3)).cuda()
            col = torch.rand(B, N, 3, generator=g).cuda()
            lab = torch.randint(0, 5, (B, N), generator=g).cuda()
            hits0 = rowmlp._step_stats[0]
            for net, opt, prep in ((a, oa, True), (b, ob, False)):
                rowmlp.set_step_operands(prep)  # network b: every stack prepares its own operands, as before
                torch.manual_seed(100 + step)   # FPS start indices: the same CPU-generator draw for both
                opt.zero_grad(set_to_none=True)
                F.cross_entropy(net(xyz, col), lab).backward()
                opt.step()
                if prep:
                    rowmlp.prepare_step()
            if step == 0:
                assert rowmlp._step_stats[0] == hits0  # first step: the stacks only registered
            else:
                assert rowmlp._step_stats[0] - hits0 >= 15, rowmlp._step_stats  # later steps: found prepared
        rowmlp.set_step_operands(True)
        # the two trajectories stay together as far as Adam lets two runs of one command (scatter-adds by atomics,
        # sign-like first updates: see test_bench_two_ranks_strong_scaling_equal_the_one_rank_run)
        a.eval(), b.eval()
        with torch.no_grad():
            torch.manual_seed(9)
            la = a(xyz, col)
            torch.manual_seed(9)
            lb = b(xyz, col)
        assert float((la - lb).abs().max() / lb.abs().max()) < (3e-2 if precision == "fp32" else 2.5e-1)
        # same weights in both: a forward pass on prepared operands IS the forward pass on self-prepared ones
        b.load_state_dict(a.state_dict())
        rowmlp.load_centre_state(b, rowmlp.centre_state(a))   # engine state beside the state_dict (bf16 rows)
        a.train(), b.train()
        rowmlp.prepare_step()
        hits = rowmlp._step_stats[0]
        torch.manual_seed(10)
        ya = a(xyz, col)
        assert rowmlp._step_stats[0] - hits >= 15
        rowmlp.set_step_operands(False)
        torch.manual_seed(10)
        yb = b(xyz, col)
        assert torch.equal(ya, yb)
        # a weight edited after prepare_step(): the stale operands must not be used
        rowmlp.set_step_operands(True)
        b.load_state_dict(a.state_dict())   # (running statistics moved in the forward passes above)
        rowmlp.load_centre_state(b, rowmlp.centre_state(a))
        rowmlp.prepare_step()
        with torch.no_grad():
            for net in (a, b):
                net.sa1.conv_blocks[0][0].weight.mul_(1.5)
        torch.manual_seed(11)
        ya = a(xyz, col)
        rowmlp.set_step_operands(False)
        torch.manual_seed(11)
        yb = b(xyz, col)
        assert torch.equal(ya, yb)
    finally:
        rowmlp.set_step_operands(True)
        rowmlp.set_precision("fp32")


def test_captured_inference_pass_equals_the_eager_pass(monkeypatch):
    """bench.py --mode infer --exec graph: the eval forward pass of a batch replayed from one hipGraph, the NEXT batch's
    sampling pyramid / ball queries / k-NN computed inside it on a forked stream into the staging set the next replay
    commits.  With the FPS start indices pinned (torch.randint -> zeros) every replay's logits must equal the plain
    eager pass over the same batch bit for bit -- three consecutive batches, i.e. two hand-overs of the double set."""
    import argparse
    import bench
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models import pointnet2_utils as pu
    real_randint = torch.randint

    def zeros_on_cpu(*a, **k):
        out = real_randint(*a, **k)
        return out.zero_() if out.device.type == "cpu" else out

    monkeypatch.setattr(torch, "randint", zeros_on_cpu)
    args = argparse.Namespace(no_dropout=True, no_prefetch=False, dump=False)
    run = bench.Run(args, "pn2_msg", "bf16", 4, 4096, 0, 1, torch.device("cuda", 0), mode="infer", graph=True)
    try:
        for step in range(3):
            batch = run._batch()
            run.infer_step()
            torch.cuda.synchronize()
            got = run.logits_out.clone()
            pu.set_static_sampling(None)
            with torch.no_grad():
                want = run.model(batch[0], batch[1])
            pu.set_static_sampling(run.static)
            assert torch.equal(got, want), (step, float((got - want).abs().max()))
    finally:
        run.close()
        pu.set_static_sampling(None)
        rowmlp.set_precision("fp32")


def test_copy_table_moves_every_buffer():
    """pcb_copy_table: many device-to-device copies from one launch (the captured steps' staging -> live hand-over):
    buffers of different dtypes and sizes, one of them not 16-byte aligned, bytes beyond each buffer untouched."""
    from pointcloud_bridge_amd import _lib
    lib = _lib.load()
    torch.manual_seed(8)
    srcs = [torch.randint(0, 1 << 40, (16, 1024), device="cuda"), torch.randn(16, 512, 3, device="cuda"),
            torch.randn(100003, device="cuda")[1:], torch.randint(0, 100, (7,), device="cuda", dtype=torch.int32),
            torch.randn(5_000_000, device="cuda")]
    dsts = [torch.full_like(t, 3) for t in srcs]
    guard = [torch.full((t.numel() + 8,), 7, dtype=t.dtype, device="cuda") for t in srcs]   # copies land in the middle
    dsts = [g[4:4 + t.numel()].view(t.shape) for g, t in zip(guard, srcs)]
    vals, blocks = [], 0
    for d, r in zip(dsts, srcs):
        nbytes = r.numel() * r.element_size()
        vals += [d.data_ptr(), r.data_ptr(), nbytes, blocks]
        blocks += (nbytes + 16383) // 16384
    table = torch.tensor(vals, dtype=torch.int64, device="cuda")
    assert lib.pcb_copy_table(table.data_ptr(), len(srcs), blocks, _stream()) == 0
    torch.cuda.synchronize()
    for g, d, r in zip(guard, dsts, srcs):
        assert torch.equal(d, r)
        assert bool((g[:4] == 7).all()) and bool((g[-4:] == 7).all())


def test_conv_rows_with_the_other_branch_added_in_the_epilogue():
    """conv_rows(conv, x, add=y): y = the other branch's bf16 rows, added to the rounded GEMM result in the epilogue
    (EnhancedFeaturePropagation: trunk + boundary term, reference :296).  Equal to the separate addition, gradients
    included (d/d(add) is the identity)."""
    from pointcloud_bridge_amd import rowmlp
    with rowmlp.precision("bf16"):
        torch.manual_seed(12)
        conv = torch.nn.Conv1d(64, 128, 1).cuda()
        x = torch.randn(5000, 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
        y = torch.randn(5000, 128, device="cuda").to(torch.bfloat16).requires_grad_(True)
        g = torch.randn(5000, 128, device="cuda").to(torch.bfloat16)
        out = rowmlp.conv_rows(conv, x, add=y)
        out.backward(g)
        got = (out.detach().clone(), x.grad.clone(), y.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
        x.grad = y.grad = conv.weight.grad = conv.bias.grad = None
        ref = rowmlp.conv_rows(conv, x) + y
        ref.backward(g)
        assert torch.equal(got[0], ref.detach())          # bf16(bf16(x W^T + b) + y) both ways
        assert torch.equal(got[1], x.grad) and torch.equal(got[2], y.grad) and torch.equal(got[2], g)
        assert torch.equal(got[3], conv.weight.grad) and torch.equal(got[4], conv.bias.grad)


@pytest.mark.parametrize("two", [True, False])
def test_gemm_with_repeated_addends_and_their_gradient_sums(two):
    """pcb_gemm_nt_stats_add_bf16: out = bf16(a W^T + add1[r >> sh1] + add2[r >> sh2]) + the statistics slabs of out;
    pcb_dy_repeat_sums_bf16: sums of dy (as pcb_dy_rows_bf16 writes it) over the rows each coarse row stood for."""
    from pointcloud_bridge_amd import _lib
    L = _lib.load()
    torch.manual_seed(5)
    R, N, K, sh1, sh2 = 128 * 37, 136, 72, 2, 5     # a ragged last column tile, a partly filled last grid pass
    a = torch.randn(R, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.2).to(torch.bfloat16)
    add1 = torch.randn(R >> sh1, N, device="cuda")
    add2 = torch.randn(R >> sh2, N, device="cuda") if two else None
    out = torch.empty(R, N, dtype=torch.bfloat16, device="cuda")
    nparts = 19
    sums = torch.full((nparts, 2, N), float("nan"), device="cuda")
    assert L.pcb_gemm_nt_stats_add_bf16(a.data_ptr(), w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr(), nparts,
                                        add1.data_ptr(), sh1, 0 if add2 is None else add2.data_ptr(), sh2, 0, _stream()) == 0
    ref = a.float() @ w.float().t() + add1.repeat_interleave(1 << sh1, dim=0)
    if two:
        ref = ref + add2.repeat_interleave(1 << sh2, dim=0)
    assert (out.float() - ref).abs().max() <= 2e-2 * ref.abs().max()     # one bf16 rounding of an fp32 sum
    tot = sums.double().sum(0)
    assert torch.allclose(tot[0], out.double().sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(tot[1], (out.double() ** 2).sum(0), rtol=1e-5, atol=1e-3)
    # rows stored centred: out = bf16(product + addends - centre), statistics of those rows
    centre = torch.randn(N, device="cuda") * 3
    assert L.pcb_gemm_nt_stats_add_bf16(a.data_ptr(), w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr(), nparts,
                                        add1.data_ptr(), sh1, 0 if add2 is None else add2.data_ptr(), sh2,
                                        centre.data_ptr(), _stream()) == 0
    refc = ref - centre
    assert (out.float() - refc).abs().max() <= 2e-2 * refc.abs().max()
    tot = sums.double().sum(0)
    assert torch.allclose(tot[0], out.double().sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(tot[1], (out.double() ** 2).sum(0), rtol=1e-5, atol=1e-3)
    # shifts below 2 are refused (the epilogue takes one coarse row per run of 4 output rows)
    assert L.pcb_gemm_nt_stats_add_bf16(a.data_ptr(), w.data_ptr(), R, N, K, out.data_ptr(), sums.data_ptr(), nparts,
                                        add1.data_ptr(), 1, 0, 1, 0, _stream()) < 0

    C = N
    dz = torch.randn(R, C, device="cuda").to(torch.bfloat16)
    y = torch.randn(R, C, device="cuda").to(torch.bfloat16)
    scale, shift, p, q = (torch.randn(C, device="cuda") * s for s in (1.0, 0.3, 0.05, 0.05))
    dy = torch.empty(R, C, dtype=torch.bfloat16, device="cuda")
    assert L.pcb_dy_rows_bf16(dz.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), p.data_ptr(), q.data_ptr(),
                              1, R, C, dy.data_ptr(), _stream()) == 0
    d1 = torch.full((R >> sh1, C), float("nan"), device="cuda")
    d2 = torch.full((R >> sh2, C), float("nan"), device="cuda") if two else None
    assert L.pcb_dy_repeat_sums_bf16(dz.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), p.data_ptr(),
                                     q.data_ptr(), 1, R, C, sh1, d1.data_ptr(), sh2, 0 if d2 is None else d2.data_ptr(),
                                     _stream()) == 0
    assert torch.allclose(d1, dy.float().view(R >> sh1, 1 << sh1, C).sum(1), rtol=1e-5, atol=1e-5)
    if two:
        assert torch.allclose(d2, dy.float().view(R >> sh2, 1 << sh2, C).sum(1), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("train", [True, False])
def test_conv_over_repeated_levels_equals_the_concatenated_form(train):
    """rowmlp.conv_bn_act_levels (MultiScaleFeatureFusion's upsample + concat followed by final_fusion's first
    conv + BatchNorm, models/model.py:150-170, :93-99, without the concatenated rows) against the same layer on
    repeat_concat's rows: output, running statistics and every gradient."""
    from pointcloud_bridge_amd import rowmlp
    with rowmlp.precision("bf16"):
        torch.manual_seed(21)
        R, reps, widths = 4096, [32, 16, 1], [128, 64, 128]
        conv = torch.nn.Conv1d(sum(widths), 128, 1).cuda()
        bn = torch.nn.BatchNorm1d(128).cuda()
        bn.train(train)
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
        ref_bn = torch.nn.BatchNorm1d(128).cuda()
        ref_bn.load_state_dict(bn.state_dict())
        ref_bn.train(train)
        levels = [torch.randn(R // r, c, device="cuda").to(torch.bfloat16).requires_grad_(True) for r, c in zip(reps, widths)]
        g = torch.randn(R, 128, device="cuda").to(torch.bfloat16)

        out = rowmlp.conv_bn_act_levels(conv, bn, levels, reps)
        out.backward(g)
        got = [out.detach().float()] + [t.grad.float() for t in levels] + [conv.weight.grad.clone(), bn.weight.grad.clone(),
                                                                            bn.bias.grad.clone()]
        for t in levels + [conv.weight, conv.bias, bn.weight, bn.bias]:
            t.grad = None
        ref = rowmlp.conv_bn_act(conv, ref_bn, rowmlp.repeat_concat(levels, reps))
        ref.backward(g)
        want = [ref.detach().float()] + [t.grad.float() for t in levels] + [conv.weight.grad, ref_bn.weight.grad, ref_bn.bias.grad]
        names = ["out", "d level0", "d level1", "d level2", "d weight", "d gamma", "d beta"]
        for n, a, b in zip(names, got, want):
            err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
            assert err < 2e-2, (n, err)     # bf16 operands / results on both sides, fp32 sums in a different order
        assert torch.allclose(bn.running_mean, ref_bn.running_mean, rtol=1e-3, atol=1e-3)
        assert torch.allclose(bn.running_var, ref_bn.running_var, rtol=1e-3, atol=1e-3)
        # fp32 rows: the concatenated form runs (no fused variant), same call
        with rowmlp.precision("fp32"):
            out32 = rowmlp.conv_bn_act_levels(conv, bn, [t.detach().float() for t in levels], reps)
        assert out32.dtype == torch.float32 and out32.shape == (R, 128)
