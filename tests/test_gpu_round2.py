"""GPU: parity cases added in round 2.

  * sample_and_group against the reference's own output (direct golden)
  * the reference training loop's losses over several Adam steps (row H1)
  * bf16 networks against the REFERENCE fixtures, with bars set from measured errors
  * a captured (hipGraph) DGCNN step against the eager step
  * the eval-mode operand cache under FlatAdam steps and library-side running-statistics updates
  * prefetched results are not inherited by a new tensor that lands at a dropped batch's address
  * SyncBatchNorm in the fused engine: two ranks == one rank over the global batch (both row types)
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.helpers import free_port, load_golden
from tests.test_gpu_modules import assert_grad_norms, build, dev, dropout_eval, grad_norms, rel_err, run_seg

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag", ["cont", "grid"])
def test_sample_and_group_golden(tag):
    """sample_and_group (reference :42-60): FPS + ball query + grouping; new_xyz bitwise, new_points exact
    (coordinates first, then the features), with and without features; same CPU-generator draw."""
    from pointcloud_bridge_amd.models import pointnet2_utils as mpu
    g = load_golden("sample_and_group")
    xyz, pts = dev(g[f"{tag}_xyz"]), dev(g["points"])
    args = (int(g["npoint"]), float(g["radius"]), int(g["nsample"]))
    torch.manual_seed(int(g["fwd_seed"]))
    new_xyz, new_points = mpu.sample_and_group(*args, xyz, pts)
    assert np.array_equal(new_xyz.cpu().numpy(), g[f"{tag}_new_xyz"])
    assert np.array_equal(new_points.cpu().numpy(), g[f"{tag}_new_points"])
    torch.manual_seed(int(g["fwd_seed"]))
    new_xyz0, new_points0 = mpu.sample_and_group(*args, xyz, None)
    assert np.array_equal(new_xyz0.cpu().numpy(), g[f"{tag}_new_xyz"])
    assert np.array_equal(new_points0.cpu().numpy(), g[f"{tag}_new_points_nofeat"])


def test_training_steps_follow_the_reference_loop():
    """Row H1: the reference's PointNet2 under the reference loop's Adam + CrossEntropy for 4 steps over
    two batches (tests/golden/make_golden_round2.py).  The build's network on the HIP path (fp32 rows)
    must follow the same trajectory.  The first loss (no update yet) matches to 1e-5.  From there on
    Adam's first updates are sign-like -- lr * g / (|g| + 1e-8) -- so entries whose gradient is
    rounding noise (1e-8 and below: e.g. the conv biases in front of a BatchNorm, exactly 0.0 here and
    1e-9-sized noise in the reference) move by up to lr in directions that differ between any two
    implementations; measured on MI355X: 1.1e-4, 1.2e-3, 1.8e-3 relative on the losses of steps 2-4.
    Bars: 5e-3 on the losses, 5e-2 of the logit range after the last step."""
    from pointcloud_bridge_amd.models.containers import PointNet2
    g = load_golden("train_steps")
    model = build(PointNet2, g["init_seed"], 5)
    model.train()
    dropout_eval(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    batches = [(dev(g[f"xyz{i}"]), dev(g[f"colors{i}"]), dev(g[f"labels{i}"])) for i in range(2)]
    torch.manual_seed(int(g["fwd_seed"]))
    losses = []
    for i in range(int(g["steps"])):
        xyz, colors, labels = batches[i % 2]
        opt.zero_grad()
        loss = F.cross_entropy(model(xyz, colors), labels)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    print("losses", losses, "reference", g["losses"].tolist())
    assert abs(losses[0] - g["losses"][0]) < 1e-5 * g["losses"][0]
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-3)
    assert losses[-1] < losses[0] - 0.1   # it learns, as the reference does (1.626 -> 1.442)
    model.eval()
    with torch.no_grad():
        final = model(batches[0][0], batches[0][1])
    print("final logits rel err", rel_err(final, g["final_logits_eval"]))
    assert rel_err(final, g["final_logits_eval"]) < 5e-2


# Measured on MI355X (tools/bf16_parity.py, round 3: rows stored centred): bf16 rows against the REFERENCE's fp32 outputs
# (relative: max |d| / max |ref| and mean |d| / mean |ref| for logits).
#   name            eval max / mean       train max / mean      loss      grad norms median / max
#   pn2_ssg         4.7e-3 / 1.6e-3       1.55e-1 / 1.43e-1     4.9e-4    1.3e-2 / 0.17      (round 2: 1.48e-1 / 1.64e-1, 2.1e-2 / 0.25)
#   pn2_ssg_skip    5.2e-3 / 2.0e-3       9.0e-2  / 1.27e-1     2.6e-4    1.7e-2 / 0.21      (         1.08e-1 / 1.46e-1, 1.6e-2 / 0.20)
#   pn2_msg         5.9e-3 / 1.9e-3       1.58e-1 / 1.12e-1     5.0e-4    2.1e-2 / 0.21      (         1.53e-1 / 1.34e-1, 2.1e-2 / 0.31)
# Eval mode is bf16 rounding noise (2^-9 per stored activation).  Train mode is ~50x worse, and round 2 blamed the wrong
# thing (2^-9 |y| divided by std(y), i.e. the |mean|/std of the stored rows): centred storage removes exactly that factor
# and buys 15 % of the mean error.  What tools/bf16_mixed.py measures instead: a freshly initialised ReLU + BatchNorm
# network AMPLIFIES independent noise relative to its signal, ~1.2x per layer -- a ReLU halves the variance of the
# noise but turns two thirds of the signal's variance into a mean that the next train-mode BatchNorm removes (eval-mode
# BatchNorm of a fresh network removes nothing, hence the 50x).  ONE perturbation of relative size 2^-9 behind sa1 of the
# fp32 engine arrives at the logits as 3.0e-2 (max) / 2.1e-2 (mean); behind sa2 1.8e-2, sa3 5.6e-3, fp1 1.2e-3.  With ~9
# roundings per stage the bf16 engine cannot be closer than ~1e-1 to fp32 train-mode logits of THIS network at
# initialisation whatever it stores (fp32 rows up to and including sa1: 5e-2; sa1+sa2: 2.5e-2; the whole encoder:
# 2.0e-2) -- and the fp32 engine's own 6e-6 is the same amplification applied to 2^-24.  The loss and the gradient norms
# stay close because the error is a smooth per-point perturbation.  rowmlp.bind_precision(model.sa1, "fp32") is the knob
# for a caller who wants the first stages exact.  Bars = about twice the measured values.
_BF16_BARS = {
    "model_pn2_ssg": dict(eval_max=1.2e-2, eval_mean=5e-3, train_max=0.3, train_mean=0.3, loss=3e-3, gn_median=4.5e-2, gn_max=0.45),
    "model_pn2_ssg_skip": dict(eval_max=1.2e-2, eval_mean=5e-3, train_max=0.3, train_mean=0.3, loss=3e-3, gn_median=4.5e-2, gn_max=0.45),
    "model_pn2_msg": dict(eval_max=1.2e-2, eval_mean=5e-3, train_max=0.3, train_mean=0.3, loss=3e-3, gn_median=4.5e-2, gn_max=0.45),
}


def _bf16_errors(name, kw):
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models import containers
    g = load_golden(name)
    kw = dict(kw)
    cls = getattr(containers, kw.pop("cls"))
    model = build(cls, g["init_seed"], 5, **kw)
    with rowmlp.precision("bf16"):
        le, lt, loss = run_seg(model, dev(g["xyz"]), dev(g["colors"]), dev(g["labels"]), int(g["fwd_seed"]), 1)
    out = {}
    for tag, got, ref in (("eval", le, g["logits_eval"]), ("train", lt, g["logits_train"])):
        d = np.abs(got.float().detach().cpu().numpy() - ref)
        out[f"{tag}_max"] = float(d.max() / np.abs(ref).max())
        out[f"{tag}_mean"] = float(d.mean() / np.abs(ref).mean())
    out["loss"] = abs(loss - float(g["loss"])) / abs(float(g["loss"]))
    gn, ref = grad_norms(model), g["grad_norms"]
    big = ref > 1e-3 * ref.max()   # biases in front of a BatchNorm: zero in exact arithmetic
    r = np.abs(gn[big] - ref[big]) / ref[big]
    out["gn_median"], out["gn_max"] = float(np.median(r)), float(r.max())
    return out


@pytest.mark.parametrize("name,kw", [
    ("model_pn2_ssg", dict(cls="PointNet2", rgb_skip=False)),
    ("model_pn2_ssg_skip", dict(cls="PointNet2", rgb_skip=True)),
    ("model_pn2_msg", dict(cls="PointNet2MSG")),
])
def test_bf16_networks_against_reference_fixtures(name, kw):
    """The arithmetic bench.py times (bf16 rows) against the reference's own fp32 outputs: eval and
    train logits, loss, per-parameter gradient norms.  Same sampling (fp32 geometry), so the differences
    are bf16 rounding of the activations only."""
    e = _bf16_errors(name, kw)
    bars = _BF16_BARS[name]
    print(name, {k: f"{v:.3e}" for k, v in e.items()})
    for key, bar in bars.items():
        assert e[key] < bar, (key, e[key], bar)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 8e-2)])
def test_dgcnn_captured_step_equals_eager_step(precision, tol):
    """A DGCNN training step (forward + CrossEntropy + backward: grid kNN + three feature-space kNN
    graphs, EdgeConv blocks, global max-pool, head) captured into ONE hipGraph must give the eager
    step's loss and gradients -- on a first replay AND on replays that follow ordinary host-side
    allocator activity.  Round 1 refused --graph for DGCNN after a replay ended in a GPU memory fault.
    The fault is reproducible (tools/dgcnn_repro.py): back-to-back replays are clean; a replay after
    ordinary tensor allocations died in ATen's scatter kernel -- the backward of the
    `x.max(dim=1)` global pool, the only consumer of an UNCLAMPED index tensor in the step (every own
    kernel clamps its indices).  The pool is now csrc/scenepool.hip (int32 row indices compared, never
    dereferenced).  Tolerances: fp32 rows 2e-4 (fp32 atomics reorder sums); bf16 rows 8e-2 -- two EAGER
    bf16 steps differ by 2e-2 on the BatchNorm gradients of the first blocks (atomic order -> bf16
    rounding of the scattered gradients), measured with the same tool."""
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    g = load_golden("model_dgcnn")
    xyz, colors, labels = dev(g["xyz"]), dev(g["colors"]), dev(g["labels"])
    model = build(DGCNN, g["init_seed"], 5, k=20).train()
    with rowmlp.precision(precision):
        def step():
            loss = F.cross_entropy(model(xyz, colors).reshape(-1, 5), labels.reshape(-1))
            loss.backward()
            return loss

        for p in model.parameters():
            p.grad = None
        eager_loss = float(step())
        eager = [p.grad.clone() for p in model.parameters()]
        gmax = max(float(t.abs().max()) for t in eager)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for p in model.parameters():
                p.grad = None
            step()   # warm-up on a side stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in model.parameters():
            p.grad = None
        loss_buf = torch.zeros((), device="cuda")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss_buf.copy_(step().detach())
        for replay in range(3):
            graph.replay()
            torch.cuda.synchronize()
            assert abs(float(loss_buf) - eager_loss) < 1e-3 * abs(eager_loss)
            for a, b in zip([p.grad for p in model.parameters()], eager):
                # the comparison itself is the host-side allocator activity in front of the next replay
                # (gradients that are zero in exact arithmetic -- bn5.bias behind the pool and the next
                # BatchNorm -- are rounding noise on both sides: floor at 1 % of the largest gradient)
                assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-2 * gmax), replay
            junk = [torch.randn(257, 1031, device="cuda") for _ in range(8)]  # more churn: fresh blocks, written
            del junk


def test_eval_cache_sees_flat_adam_steps_and_running_stat_updates():
    """the eval-mode operand cache must not serve stale weights after parallel.FlatAdam
    stepped the flat buffer the parameters are views of, nor stale BatchNorm constants after a
    training-mode forward updated the running statistics inside the library (neither bumps a tensor
    version counter)."""
    from pointcloud_bridge_amd import parallel, rowmlp
    from pointcloud_bridge_amd.models.containers import PointNet2
    torch.manual_seed(1)
    model = PointNet2(5).cuda()
    B, N = 2, 1024
    xyz = torch.rand(B, N, 3, device="cuda")
    col = torch.rand(B, N, 3, device="cuda")
    lab = torch.randint(0, 5, (B, N), device="cuda")
    params = [p for p in model.parameters() if p.requires_grad]
    opt = parallel.FlatAdam(params, lr=1e-2)

    def evaluate():
        model.eval()
        with torch.no_grad():
            torch.manual_seed(5)
            return model(xyz, col).float().clone()

    for precision in ("bf16", "fp32"):
        with rowmlp.precision(precision):
            first = evaluate()
            assert torch.equal(first, evaluate())           # cached operands: identical
            model.train()
            for p in params:
                p.grad = None
            torch.manual_seed(5)
            F.cross_entropy(model(xyz, col), lab).backward()  # updates running statistics in the library
            opt.step()                                        # updates the parameters through the flat buffer
            after = evaluate()
            assert not torch.equal(first, after), "eval output unchanged after a training step: stale cache"
            rowmlp._eval_operands.clear()
            assert torch.equal(after, evaluate()), "cached eval differs from a freshly prepared one"


def test_prefetch_is_not_inherited_by_a_new_tensor_at_the_same_address():
    """prefetch(A); A is dropped without being forwarded; B (same shape) lands at
    A's address.  B's forward must equal the non-prefetched result -- for the PointNet++ pyramid, the
    DGCNN coordinate graph and the bridge geometry."""
    from pointcloud_bridge_amd.models import pointnet2_utils as mpu
    from pointcloud_bridge_amd.models.containers import EnhancedPointNet2, PointNet2MSG
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    torch.manual_seed(2)
    B, N = 2, 1100   # an unusual size: the freed block is the allocator's exact fit for the next request
    col = torch.rand(B, N, 3, device="cuda")
    for make in (lambda: PointNet2MSG(5), lambda: DGCNN(5, k=8), lambda: EnhancedPointNet2(5)):
        torch.manual_seed(3)
        model = make().cuda().eval()
        torch.cuda.synchronize()
        a = torch.rand(B, N, 3, device="cuda")
        ptr = a.data_ptr()
        with torch.no_grad():
            torch.manual_seed(9)
            model.prefetch(a)
            torch.cuda.synchronize()
            del a                                   # dropped without a forward pass
            b = torch.rand(B, N, 3, device="cuda")  # the caching allocator hands out A's block again
            if b.data_ptr() != ptr:
                pytest.skip("allocator did not reuse the block: the hazard cannot be staged here")
            torch.manual_seed(11)
            got = model(b, col)
            if hasattr(model, "sampling"):          # (DGCNN parks its coordinate graph on the module itself)
                model.sampling.prefetched.clear()
                model.sampling.parked.clear()
            mpu._default_state.prefetched.clear()
            mpu._default_state.parked.clear()
            torch.manual_seed(11)
            want = model(b, col)
        assert torch.equal(got, want), type(model).__name__


def test_syncbatchnorm_two_ranks_equal_one_rank_global_batch(tmp_path):
    """SyncBatchNorm in the fused engine.  Two ranks (gloo, sharing this GPU; the launcher
    is a fresh child process started before anything here touches the device in it) each run a stack
    on half of the rows; rank 0 compares with the single-process stack over all rows."""
    out = tmp_path / "syncbn.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PCB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(REPO, "tests", "_syncbn_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = out.read_text().strip().splitlines()
    print("\n".join(lines))
    assert len(lines) == 6 and all(": OK" in ln for ln in lines), lines


def test_bench_two_ranks_strong_scaling_equal_the_one_rank_run(tmp_path):
    """SURVEY 8(e) end to end through bench.py: ONE global batch of 4 scenes, (a) in one process, (b) split
    over two ranks (gloo; both on this GPU; the launcher is a fresh child process) with SyncBatchNorm
    inside the fused engine, FPS start indices drawn for the global batch on every rank
    (pointnet2_utils.set_scene_shard) and the overlapped flat gradient all-reduce.  Same data, same
    seeds, fp32 rows.  The first step's loss agrees to 1e-6 and its averaged gradient equals the
    single-process gradient up to the branches ReLU / max-pool take at pre-activations within rounding
    distance of a tie (the all-reduced statistics differ from the single process's in the last bit;
    two identical single-process runs agree to 2e-6, tools/grad_noise.py; measured here: 99 % quantile
    3.1e-4, relative L2 6.1e-3, worst entry 6.2e-3): 99 % of the entries within 1e-3 of the largest,
    relative L2 error below 2e-2.  The global mean losses of 3 steps agree to 1e-3
    and the parameters stay within the reach of Adam's sign-like first updates."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PCB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "3", "--warmup", "0", "--npoints", "2048", "--batch", "4", "--precision", "fp32",
              "--no-cpu-baseline", "--no-extras", "--no-dropout", "--model", "pn2_msg"]
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--dump", one] + common,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2",
                        "--scaling", "strong", "--sync-bn", "--dump", two] + common,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    a, b = torch.load(one, weights_only=True), torch.load(two, weights_only=True)
    print("losses", a["losses"], b["losses"])
    np.testing.assert_allclose(b["losses"], a["losses"], rtol=1e-3)
    ga, gb = a["first_grad"], b["first_grad"]
    assert ga.numel() == gb.numel() > 0
    gerr = float((ga - gb).abs().max() / ga.abs().max())
    d = (ga - gb).abs()
    q99 = float(d.kthvalue(int(d.numel() * 0.99))[0] / ga.abs().max())
    l2 = float((ga - gb).norm() / ga.norm())
    print("first-step gradient: max |d| / max |g| =", gerr, " 99 % quantile", q99, " relative L2", l2)
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    off, rows = 0, []
    for name, prm in PointNet2MSG(5).named_parameters():
        n = prm.numel()
        rows.append((float((ga[off:off + n] - gb[off:off + n]).abs().max() / ga.abs().max()), name))
        off += n
    print("worst parameters:", sorted(rows, reverse=True)[:6])
    for e, name in rows:
        if name.endswith("weight") and ("convs" in name or "conv_blocks" in name or "final" in name or "fusion" in name or "attention" in name):
            print(f"   {name:40s} {e:.2e}")
    assert abs(a["losses"][0] - b["losses"][0]) < 1e-6 * abs(a["losses"][0])
    assert q99 < 1e-3 and l2 < 2e-2 and gerr < 0.1
    d = (a["flat"] - b["flat"]).abs()
    print("params: mean |d|", float(d.mean()), "max |d|", float(d.max()))
    assert float(d.mean()) < 1e-3 and float(d.max()) < 6.1e-3   # 3 steps x lr (1e-3) x 2 at most for a sign flip


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_repeat_concat_equals_expand_cat(precision):
    """rowmlp.repeat_concat (MultiScaleFeatureFusion's nearest upsampling + cat, models/model.py:150-170) against
    expand + torch.cat: forward bit-exact; backward = the block of the gradient summed over the repeats in fp32
    (bf16 rows: within one rounding of the fp32 sum)."""
    from pointcloud_bridge_amd import rowmlp
    with rowmlp.precision(precision):
        m = rowmlp.mode()
        torch.manual_seed(5)
        B, n = 3, 64
        reps, widths = [16, 4, 1], [16, 24, 8]
        levels = [torch.randn(B * n // r, c, device="cuda").to(m.dtype).requires_grad_(True) for r, c in zip(reps, widths)]
        out = rowmlp.repeat_concat(levels, reps)
        ref_levels = [t.detach().float().requires_grad_(True) for t in levels]
        ref = torch.cat([t.view(B * n // r, 1, c).expand(B * n // r, r, c).reshape(B * n, c)
                         for t, r, c in zip(ref_levels, reps, widths)], dim=1)
        assert out.shape == ref.shape and torch.equal(out.float(), ref)
        g = torch.randn_like(ref).to(m.dtype)
        out.backward(g)
        ref.backward(g.float())
        tol = 1e-6 if precision == "fp32" else 2.0 ** -8
        for t, r in zip(levels, ref_levels):
            assert t.grad.shape == r.grad.shape
            err = (t.grad.float() - r.grad).abs().max() / r.grad.abs().max()
            assert err < tol, err


@pytest.mark.gpu
def test_split_cols_backward_is_the_concatenation():
    from pointcloud_bridge_amd import rowmlp
    w = torch.randn(12, 19, device="cuda", requires_grad=True)
    a, b = rowmlp.split_cols(w, 3)
    assert a.data_ptr() == w.data_ptr() and a.stride() == (19, 1) and b.stride() == (19, 1)   # views: consumers read them in place
    assert torch.equal(a, w[:, :3]) and torch.equal(b, w[:, 3:])
    (a.sum() * 2 + (b * b).sum()).backward()
    ref = torch.cat([torch.full((12, 3), 2.0, device="cuda"), 2 * w.detach()[:, 3:]], dim=1)
    assert torch.equal(w.grad, ref)
    w.grad = None
    a, b = rowmlp.split_cols(w, 3)
    b.sum().backward()   # one piece unused: its block of the gradient is zero
    assert torch.equal(w.grad[:, :3], torch.zeros(12, 3, device="cuda")) and torch.equal(w.grad[:, 3:], torch.ones(12, 16, device="cuda"))


@pytest.mark.gpu
def test_branch_streams_change_nothing():
    """The independent sub-chains of PN2-MSG (MSG scales, decoder boundary terms, fusion levels) on streams of
    their own (pointnet2_utils.run_branches) against the same step on one stream: the same kernels in the same
    per-chain order.  The step itself is not bit-reproducible (the gathered set abstraction scatters du with fp32
    atomics, and one changed bf16 rounding moves a ReLU mask), so the yardstick is the spread between two
    single-stream runs: the branch-stream runs must stay within a few times that.  A missing synchronisation
    (forward join, saved inputs released under a running branch, gradient buckets packed during backward) shows
    as errors of the gradients' own size -- test_bridgeseg_logit_parity caught one at 30 %."""
    import bench
    from pointcloud_bridge_amd import parallel, rowmlp
    from pointcloud_bridge_amd.models import pointnet2_utils as pu
    rowmlp.set_precision("bf16")
    try:
        dev = torch.device("cuda", 0)
        torch.manual_seed(42)
        model, cdim = bench.build_model("pn2_msg")
        model = model.to(dev).train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        state = {k: v.clone() for k, v in model.state_dict().items()}
        xyz, colors, labels = bench.synthetic_batch(4, 4096, 123, dev)
        bucket = parallel.OverlappedGradAllReduce.by_children(model)

        def step(flag):
            pu.set_branch_streams(flag)
            model.load_state_dict(state)
            rowmlp.reset_centres(model)
            torch.manual_seed(7)  # FPS start indices
            bucket.zero()
            loss = bench.loss_fn(model(xyz, colors), labels, cdim)
            loss.backward()
            flat = bucket.finish().clone()
            junk = [torch.randn(1 << 20, device=dev) for _ in range(4)]  # allocator churn between runs
            del junk
            return float(loss), flat

        ref_loss, ref_flat = step(False)
        loss2, flat2 = step(False)
        scale = float(ref_flat.abs().max())
        spread = float((flat2 - ref_flat).abs().max())
        l2_spread = float((flat2 - ref_flat).norm() / ref_flat.norm())
        assert abs(loss2 - ref_loss) < 1e-5 * abs(ref_loss)
        for _ in range(3):
            loss, flat = step(True)
            assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss)
            err = float((flat - ref_flat).abs().max())
            l2 = float((flat - ref_flat).norm() / ref_flat.norm())
            assert err <= 4 * spread + 1e-6 * scale, (err, spread, scale)
            assert l2 <= 4 * l2_spread + 1e-6, (l2, l2_spread)
    finally:
        pu.set_branch_streams(True)
        rowmlp.set_precision("fp32")


@pytest.mark.gpu
def test_confusion_matrices_on_device_equal_the_per_point_loop():
    """Row f2: inference.py:226-231 fills the global and the per-file confusion matrices point by point in a
    Python loop (`for t, p in zip(targets, preds): cm[t, p] += 1`).  train.confusion_matrix /
    per_scene_confusion on DEVICE tensors (one bincount each) against that loop restated in numpy, and the
    metrics of inference.py:814-855 against their definition term by term.  (The reference module itself needs
    laspy / seaborn and cannot be imported: parity with a reference RUN stays unpinned.)"""
    from pointcloud_bridge_amd import train
    rng = np.random.default_rng(3)
    B, N, C = 5, 4096, 5
    target = rng.integers(0, C, (B, N))
    pred = np.where(rng.random((B, N)) < 0.7, target, rng.integers(0, C, (B, N)))
    pred[:, :7] = 4      # a class that is predicted but ...
    target[target == 3] = 2  # ... one that never occurs as truth
    per = np.zeros((B, C, C), np.int64)
    for b in range(B):
        for t, p in zip(target[b], pred[b]):
            per[b, t, p] += 1
    dt, dp = torch.from_numpy(target).cuda(), torch.from_numpy(pred).cuda()
    got_per = train.per_scene_confusion(dp, dt, C)
    got = train.confusion_matrix(dp, dt, C)
    assert got.is_cuda and got_per.is_cuda
    assert np.array_equal(got_per.cpu().numpy(), per)
    assert np.array_equal(got.cpu().numpy(), per.sum(0))
    m = train.metrics_from_confusion(got)
    cm = per.sum(0).astype(np.float64)
    diag = np.diag(cm)
    iou = diag / (cm.sum(1) + cm.sum(0) - diag + 1e-6)
    assert abs(float(m["miou"]) - iou.mean()) < 1e-12
    assert abs(float(m["oa"]) - diag.sum() / cm.sum()) < 1e-9
    assert float(m["iou"][3]) == 0.0   # absent class: IoU 0 (not NaN) and it DOES enter the mean (:826, :846)


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["pn2_msg", "bridgeseg"])
def test_captured_pn2_msg_step_equals_the_eager_step(model_name):
    """(bridgeseg, round 3: the reference's whole BridgeSeg network -- its colour / fusion stacks run BatchNorm on the
    narrow-row kernels and every partial-sum total goes through pcb_sum_slabs, so the step holds no ATen reduction any
    more and replays; the neighbourhood geometry of its three structure encoders rides in the static pipeline.)
    The captured hipGraph step of bench.py (static sampling pipeline: FPS pyramid, ball queries, decoder k-NN and
    inverted indices of the NEXT batch computed on a side stream into a staging set, committed at the top of the
    next replay) against the same step launched kernel by kernel: same batches in the same order, same FPS start
    indices -> the same gradients up to the run-to-run spread of the step (fp32 atomics in the gathered set
    abstraction).  With ONE set of static buffers the backward pass read centroid coordinates and neighbour indices
    that the side stream was already overwriting for the next batch."""
    import argparse
    import bench
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models import pointnet2_utils as pu
    args = argparse.Namespace(no_dropout=True, no_prefetch=False, dump=False)
    dev = torch.device("cuda", 0)
    run = bench.Run(args, model_name, "bf16", 4, 4096, 0, 1, dev, graph=True)
    try:
        run.opt.step = lambda *a, **k: None      # parameters stay put: gradients of different steps are comparable

        def pair(step):
            run.i = 0
            torch.manual_seed(11)                # FPS start indices of both steps
            step()                               # computes the pyramid of batch 1 beside its backward pass
            step()                               # batch 1: forward on that pyramid
            torch.cuda.synchronize()
            return run.bucket.flat.clone()

        e1, e2 = pair(run.eager_step), pair(run.eager_step)
        scale = float(e1.abs().max())
        spread = float((e1 - e2).abs().max())
        l2_spread = float((e1 - e2).norm() / e1.norm())
        for _ in range(2):
            g = pair(run.graph_step)
            err = float((g - e1).abs().max())
            l2 = float((g - e1).norm() / e1.norm())
            assert err <= 4 * spread + 1e-6 * scale, (err, spread, scale)
            assert l2 <= 4 * l2_spread + 1e-6, (l2, l2_spread)
    finally:
        run.close()
        pu.set_static_sampling(None)
        rowmlp.set_precision("fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("R,K,n,gap", [(4096, 16, 40, 0), (5000, 128, 5, 0), (70000, 64, 259, 3)])
def test_bf16_conv_rows_bias_gradient_from_the_weight_gradient_pass(R, K, n, gap):
    """A conv without BatchNorm in bf16 mode: dbias comes out of pcb_gemm_tn_bias_bf16 (column sums of dy kept per
    row split beside the dW slabs) -- against fp32 torch on the same bf16 operands: outputs, dx, dW, dbias."""
    from pointcloud_bridge_amd import rowmlp
    with rowmlp.precision("bf16"):
        torch.manual_seed(R + n)
        conv = torch.nn.Conv1d(K, n, 1).cuda()
        x = torch.randn(R, K, device="cuda").to(torch.bfloat16).requires_grad_(True)
        y = rowmlp.conv_rows(conv, x, out_gap=gap)
        y = rowmlp.ungap_rows(y, -gap) if gap else y
        g = torch.randn(R, n, device="cuda").to(torch.bfloat16)
        y.backward(g)
        xr = x.detach().float().requires_grad_(True)
        w = conv.weight.detach().view(n, K).to(torch.bfloat16).float().requires_grad_(True)
        b = conv.bias.detach().clone().requires_grad_(True)
        ref = xr @ w.t() + b
        ref.backward(g.float())

        def rel(a, r):
            return float((a.float() - r).abs().max() / r.abs().max())

        assert rel(y, ref.detach()) < 1e-2
        assert rel(x.grad, xr.grad) < 1e-2
        assert rel(conv.weight.grad.view(n, K), w.grad) < 2e-3
        assert rel(conv.bias.grad, b.grad) < 1e-4     # fp32 sums of the bf16 dy: only the summation order differs
