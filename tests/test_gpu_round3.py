"""GPU: round-3 cases around the engine's state.

  * precision is scoped (rowmlp.precision context, rowmlp.bind_precision per module): two networks of different row
    types interleave in one process and each gives the bits it gives alone
  * a model's operand set is its own (rowmlp.attach_step_operands): preparing, training or dropping another model
    does not touch the tables and buffers a captured step of the first one replays from
"""
import gc
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _batch(seed, B=2, N=2048):
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(B, N, 3, generator=g)
    xyz = (v / v.norm(dim=-1, keepdim=True) * torch.rand(B, N, 1, generator=g) ** (1 / 3)).cuda()
    return xyz, torch.rand(B, N, 3, generator=g).cuda(), torch.randint(0, 5, (B, N), generator=g).cuda()


def _model(seed):
    from pointcloud_bridge_amd.models.containers import PointNet2MSG
    torch.manual_seed(seed)
    m = PointNet2MSG(5).cuda().train()
    for s in m.modules():
        if isinstance(s, torch.nn.Dropout):
            s.p = 0.0
    return m


def _steps(model, batches, fps_seed):
    """Two SGD steps; returns the logits and flat gradients of each."""
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    out = []
    for i, (xyz, col, lab) in enumerate(batches):
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(fps_seed + i)
        logits = model(xyz, col)
        F.cross_entropy(logits, lab).backward()
        out.append((logits.detach().clone(), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()))
        opt.step()
    return out


def test_two_precisions_interleaved_in_one_process():
    """A bf16-bound and an fp32-bound PointNet++ MSG network take turns, step by step, inside a scope that asks for yet
    another default; each must produce exactly what it produces when it runs alone (reproducible mode, so 'exactly'
    means bit for bit).  Also: a stage bound to fp32 inside a bf16 network really runs in fp32 rows."""
    from pointcloud_bridge_amd import ops, rowmlp
    old = ops.set_deterministic(True)
    try:
        batches = [_batch(1), _batch(2)]
        alone = {}
        for prec in ("bf16", "fp32"):
            m = rowmlp.bind_precision(_model(7), prec)
            alone[prec] = _steps(m, batches, 100)
        a = rowmlp.bind_precision(_model(7), "bf16")
        b = rowmlp.bind_precision(_model(7), "fp32")
        oa, ob = torch.optim.SGD(a.parameters(), lr=1e-2), torch.optim.SGD(b.parameters(), lr=1e-2)
        got = {"bf16": [], "fp32": []}
        for i, (xyz, col, lab) in enumerate(batches):
            # (the enclosing scope's choice must not leak into either bound module -- nor theirs out of them)
            with rowmlp.precision("fp32" if i else "bf16"):
                for prec, net, opt in (("bf16", a, oa), ("fp32", b, ob)):
                    opt.zero_grad(set_to_none=True)
                    torch.manual_seed(100 + i)
                    logits = net(xyz, col)
                    assert rowmlp.get_precision() == ("fp32" if i else "bf16")
                    assert logits.dtype == torch.float32
                    F.cross_entropy(logits, lab).backward()
                    got[prec].append((logits.detach().clone(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()))
                    opt.step()
        for prec in ("bf16", "fp32"):
            for (l1, g1), (l2, g2) in zip(alone[prec], got[prec]):
                assert torch.equal(l1, l2) and torch.equal(g1, g2), prec
        assert not torch.equal(alone["bf16"][0][0], alone["fp32"][0][0])     # (they are different arithmetic)
        # mixed: the first stage in fp32 rows inside a bf16 network -- the logits move towards the fp32 network's
        mixed = rowmlp.bind_precision(_model(7), "bf16")
        rowmlp.bind_precision(mixed.sa1, "fp32")
        lm = _steps(mixed, batches[:1], 100)[0][0]
        d_mixed = float((lm - alone["fp32"][0][0]).abs().mean())
        d_bf16 = float((alone["bf16"][0][0] - alone["fp32"][0][0]).abs().mean())
        print("mean |logits - fp32 logits|: bf16", d_bf16, "bf16 with sa1 in fp32 rows", d_mixed)
        assert d_mixed < 0.8 * d_bf16
    finally:
        ops.set_deterministic(old)


def test_a_captured_step_survives_other_models_and_their_operand_sets():
    """the operand tables / buffers a captured training step replays from belong to its model's own
    StepOperands.  Capture a step of model A; then register, train, prepare and DROP a model B (and churn the
    allocator); A's replays must keep giving what an eager step of A gives."""
    from pointcloud_bridge_amd import rowmlp
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    xyz, col, lab = _batch(3, N=1024)

    def _model(seed):   # DGCNN: its step captures as it is (PointNet++ draws FPS start indices on the host: StaticSampling)
        torch.manual_seed(seed)
        return DGCNN(5, k=16).cuda().train()

    with rowmlp.precision("bf16"):
        a = _model(11)
        ops_a = rowmlp.attach_step_operands(a)
        assert rowmlp.attach_step_operands(a) is ops_a

        def step(net):
            loss = F.cross_entropy(net(xyz, col).reshape(-1, 5), lab.reshape(-1))
            loss.backward()
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):                      # warm-up: stacks register, centres settle, operands get prepared
                a.zero_grad(set_to_none=True)
                torch.manual_seed(5)
                rowmlp.prepare_step(a)
                step(a)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert len(ops_a.entries) >= 4 and ops_a.tables is not None
        hits = ops_a.stats[0]
        a.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        rowmlp.prepare_step(a)
        eager_loss = float(step(a))
        assert ops_a.stats[0] - hits >= 4           # the stacks found their operands prepared by the table launch
        eager = [p.grad.clone() for p in a.parameters()]
        for p in a.parameters():
            p.grad = None
        loss_buf = torch.zeros((), device="cuda")
        graph = torch.cuda.CUDAGraph()
        torch.manual_seed(5)
        with torch.cuda.graph(graph):
            rowmlp.prepare_step(a)
            loss_buf.copy_(step(a).detach())
        tables_before = {k: t[0].data_ptr() for k, t in ops_a.tables.items()}

        def check(tag):
            for p in a.parameters():
                p.grad.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert abs(float(loss_buf) - eager_loss) < 2e-3 * abs(eager_loss), tag
            gmax = max(float(t.abs().max()) for t in eager)
            for got, want in zip([p.grad for p in a.parameters()], eager):
                assert float((got - want).abs().max()) <= 0.1 * max(float(want.abs().max()), 1e-2 * gmax), tag

        check("first replay")
        # another model with its own set, and one in the process default set: trained, prepared, dropped
        for own in (True, False):
            b = _model(12)
            if own:
                rowmlp.attach_step_operands(b)
            opt = torch.optim.SGD(b.parameters(), lr=1e-2)
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                torch.manual_seed(6)
                step(b)
                opt.step()
                rowmlp.prepare_step(b if own else None)
            del b, opt
            gc.collect()
            torch.cuda.empty_cache()
            junk = [torch.randn(1 << 20, device="cuda") for _ in range(8)]
            del junk
            rowmlp.prepare_step(None)               # the default set prunes its dead entries: nothing of A's moves
            assert {k: t[0].data_ptr() for k, t in ops_a.tables.items()} == tables_before
            check("after another model, own set = %s" % own)


@pytest.mark.parametrize("pro,R,C,K,ns", [(2, 4096, 128, 128, 1), (2, 1000, 64, 64, 1), (3, 2048, 128, 64, 16), (3, 999 * 8, 64, 128, 8),
                                          (2, 64 * 600, 72, 40, 1), (3, 70 * 32, 128, 128, 32), (2, 50, 8, 8, 1),
                                          (3, 4096, 256, 128, 32), (2, 3000, 200, 96, 1)])
def test_fused_layer_backward_equals_the_two_kernel_form(pro, R, C, K, ns):
    """pcb_bwd_fused_bf16 (one pass over (dz, y) and the rows below: dx, dW, the sums of the layer below) against
    pcb_gemm_nt_red_bf16 + pcb_gemm_tn_bf16, the pair it replaces for the narrow layers: dx bit-identical (the same
    prologue, the same MFMA order over C), dW and the sums up to the order of the fp32 additions (other row splits)."""
    from pointcloud_bridge_amd import _lib
    from pointcloud_bridge_amd.ops import _launch
    L = _lib.load()
    assert L.pcb_bwd_fused_supported(C, K) == 1 and L.pcb_bwd_fused_supported(264, 64) == 0 and L.pcb_bwd_fused_supported(64, 136) == 0 and L.pcb_bwd_fused_supported(60, 64) == 0
    g = torch.Generator().manual_seed(R + C)
    dev = torch.device("cuda")
    y = torch.randn(R, C, generator=g).to(dev).to(torch.bfloat16)
    x = torch.randn(R, K, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(C, K, generator=g) * 0.2).to(dev).to(torch.bfloat16)          # the layer's weight [C, K]
    wt = w.t().contiguous()                                                        # prepared transposed copy [K, C]
    scale, shift = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev) * 0.3
    p, q = torch.randn(C, generator=g).to(dev) * 0.05, torch.randn(C, generator=g).to(dev) * 0.05
    xs, xh = torch.rand(K, generator=g).to(dev) + 0.5, torch.randn(K, generator=g).to(dev) * 0.3
    xm, xi = torch.randn(K, generator=g).to(dev) * 0.1, torch.rand(K, generator=g).to(dev) + 0.5
    if pro == 3:
        G = R // ns
        dout = torch.randn(G, C, generator=g).to(dev)
        arg = torch.randint(0, ns, (G, C), generator=g).to(dev).to(torch.uint8)
        dz = None
    else:
        dz = torch.randn(R, C, generator=g).to(dev).to(torch.bfloat16)
        dout = arg = None
    a_args = (0 if dz is None else dz.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), p.data_ptr(), q.data_ptr(),
              0 if dout is None else dout.data_ptr(), 0 if arg is None else arg.data_ptr(), ns, 1)
    # the two-kernel form
    nparts = L.pcb_gemm_nt_partials(pro, R, K)
    dx_ref = torch.full((R, K), float("nan"), dtype=torch.bfloat16, device=dev)
    red_ref = torch.empty(nparts, 2, K, device=dev)
    _launch("pcb_gemm_nt_red_bf16", 0, pro, *a_args, wt.data_ptr(), R, K, C, dx_ref.data_ptr(), x.data_ptr(), xs.data_ptr(),
            xh.data_ptr(), xm.data_ptr(), xi.data_ptr(), 1, red_ref.data_ptr(), nparts)
    ws = torch.empty(L.pcb_gemm_tn_workspace(R, C, K), device=dev)
    dw_ref = torch.full((C, K), float("nan"), device=dev)
    _launch("pcb_gemm_tn_bf16", 0, pro, *a_args, 1, x.data_ptr(), xs.data_ptr(), xh.data_ptr(), 1, R, C, K, ws.data_ptr(),
            dw_ref.data_ptr(), K, 0)
    # one pass
    for grid in (1, 7, min(512, (R + 63) // 64)):
        dx = torch.full((R, K), float("nan"), dtype=torch.bfloat16, device=dev)
        red = torch.full((grid, 2, K), float("nan"), device=dev)
        ws2 = torch.full((grid * C * K,), float("nan"), device=dev)
        dw = torch.full((C, K), float("nan"), device=dev)
        _launch("pcb_bwd_fused_bf16", 0, pro, *a_args, wt.data_ptr(), x.data_ptr(), xs.data_ptr(), xh.data_ptr(), xm.data_ptr(),
                xi.data_ptr(), 1, R, C, K, dx.data_ptr(), red.data_ptr(), grid, ws2.data_ptr(), dw.data_ptr(), K, 0)
        assert torch.equal(dx, dx_ref), grid
        tol = 2e-3 * float(dw_ref.abs().max())
        assert float((dw - dw_ref).abs().max()) <= tol, (grid, float((dw - dw_ref).abs().max()), tol)
        tot, tot_ref = red.double().sum(0), red_ref.double().sum(0)
        assert float((tot - tot_ref).abs().max()) <= 1e-4 * float(tot_ref.abs().max()) + 1e-4, grid


def test_cfg4_two_ranks_bridgeseg_bridge_loss_syncbn(tmp_path):
    """BASELINE configs[3] as a whole, on two ranks (gloo, both on the one GPU): the reference's BridgeSeg network
    (models/model.py:58-147) + BridgeStructureLoss (train_MulSca_BriStruNet_CB.py:151-156) + SyncBatchNorm + the
    gradient all-reduce, strong scaling (ONE global batch sharded), against the single-process run over the whole batch.
    The criterion's class weights are batch-level statistics; the global batch repeats its first half in its second
    (--dup-halves), so every rank's shard has the global batch's statistics and the two runs are the same computation:
    losses to 5e-4 at the first step, the first averaged gradient to the ReLU-flip level."""
    from tests.helpers import free_port
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PCB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "0", "--npoints", "2048", "--batch", "4", "--precision", "fp32", "--no-cpu-baseline",
              "--no-extras", "--no-dropout", "--model", "bridgeseg", "--loss", "bridge", "--dup-halves", "--scaling", "strong"]
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--dump", one] + common,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(repo, "bench.py"),
                        "--gpus", "2", "--sync-bn", "--dump", two] + common,      # (--batch = the GLOBAL batch under strong scaling)
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    a, b = torch.load(one, weights_only=True), torch.load(two, weights_only=True)
    print("losses", a["losses"], b["losses"])
    # (under SyncBatchNorm the narrow encoder layers take torch's SyncBatchNorm and the materialised neighbour rows instead
    # of the fused kernels: other fp32 summation orders in front of ~20 layers -- measured 7e-5 at the first step)
    assert abs(a["losses"][0] - b["losses"][0]) < 5e-4 * abs(a["losses"][0])
    np.testing.assert_allclose(b["losses"], a["losses"], rtol=3e-3)
    ga, gb = a["first_grad"], b["first_grad"]
    assert ga.numel() == gb.numel() > 0
    l2 = float((ga - gb).norm() / ga.norm())
    d = (ga - gb).abs()
    q99 = float(d.kthvalue(int(d.numel() * 0.99))[0] / ga.abs().max())
    print("first-step gradient: relative L2", l2, "99 % quantile", q99)
    # The duplicated scenes are sampled from their OWN farthest-point start indices (the global batch's CPU-generator
    # draw, as the reference would draw them), so a scene and its copy predict slightly different labels and the
    # criterion's per-shard class weights are close to, not equal to, the global batch's: measured 4.3e-2 relative L2 on
    # the averaged gradient.  A broken exchange (a missing all-reduce, unsynchronised statistics, a shard taking the wrong
    # scenes) shows as O(1).
    assert l2 < 0.1 and q99 < 2e-2


@pytest.mark.parametrize("model_name", ["pn2_msg", "bridgeseg"])
def test_two_segment_captured_step_equals_the_one_graph_step(model_name):
    """the captured step split into two hipGraphs at the encoder/decoder seam (bench.Run._capture_two_segments:
    the decoder's gradient bucket is all-reduced between the two replays, beside the encoder's backward segment) must
    compute what the one-graph step computes.  Run in the reproducible mode, where 'the same' means the same bits:
    losses and flat gradients of an eager step and two replays, after identical warm-up / Adam histories.  (With float
    atomics the two runs' warm-up steps already differ in the last bits, Adam's sign-like first updates amplify that
    into different weights, and nothing downstream is comparable.)  Also: the issue order of segments and exchanges."""
    import argparse
    import bench
    from pointcloud_bridge_amd import ops
    from pointcloud_bridge_amd.models import pointnet2_utils as pu
    dev = torch.device("cuda", 0)

    def flat_of(segments):
        args = argparse.Namespace(no_dropout=True, no_prefetch=False, dump=False, graph_segments=segments)
        torch.manual_seed(3)                         # the captures' own FPS draws
        run = bench.Run(args, model_name, "bf16", 4, 4096, 0, 1, dev, graph=True)
        try:
            assert getattr(run, "segments", 1) == segments
            out = []
            for step in (run.eager_step, run.graph_step, run.graph_step):
                torch.manual_seed(11)
                step()
                loss = step()
                torch.cuda.synchronize()
                out.append((float(loss), run.bucket.flat.clone(), run.opt.flat.clone()))
            if segments == 2:
                assert run.segment_log == ["A", "allreduce(decoder)", "B", "allreduce(encoder)"]
            return out
        finally:
            run.close()
            pu.set_static_sampling(None)

    old = ops.set_deterministic(True)
    try:
        one, two = flat_of(1), flat_of(2)
    finally:
        ops.set_deterministic(old)
    for (la, ga, pa), (lb, gb, pb) in zip(one, two):
        assert la == lb
        assert torch.equal(ga, gb)                   # the averaged gradient the optimiser consumed
        assert torch.equal(pa, pb)                   # ... and the parameters after it (six Adam steps so far)


def test_two_segment_step_on_a_one_rank_rccl_group(tmp_path):
    """The same on the real backend: PCB_DIST_SINGLE=1 makes bench.py join a one-rank `nccl` (= RCCL) group, so the two
    asynchronous all-reduces of the segmented step are real collectives issued between the replays.  Losses against the
    plain single-process run (one graph, no group)."""
    import json
    from tests.helpers import free_port
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "4", "--warmup", "2", "--npoints", "4096", "--batch", "4", "--no-cpu-baseline", "--no-extras",
              "--no-dropout", "--exec", "graph"]
    plain = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + common, env=env, capture_output=True, text=True,
                           timeout=600)
    assert plain.returncode == 0, plain.stdout[-2000:] + plain.stderr[-4000:]
    rccl = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(repo, "bench.py")]
                          + common + ["--graph-segments", "2"], env=dict(env, PCB_DIST_SINGLE="1"), capture_output=True,
                          text=True, timeout=600)
    assert rccl.returncode == 0, rccl.stdout[-2000:] + rccl.stderr[-4000:]
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    b = json.loads(rccl.stdout.strip().splitlines()[-1])
    assert "two graph segments" in b["config"]["workload"] and b["config"]["graph"]
    assert abs(a["config"]["loss"] - b["config"]["loss"]) < 3e-2 * abs(a["config"]["loss"])


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_dropout_rows_stateless_mask(precision):
    """rowmlp.dropout_rows (nn.Dropout(0.5) of the heads, models/model.py:97): kept elements scaled by 1/(1-p), the
    keep rate, a fresh mask per call, the SAME mask in the backward pass, identity in eval mode."""
    from pointcloud_bridge_amd import rowmlp
    with rowmlp.precision(precision):
        m = rowmlp.mode()
        drop = torch.nn.Dropout(0.5).train()
        torch.manual_seed(1)
        x = (torch.rand(4096, 128, device="cuda") + 0.5).to(m.dtype).requires_grad_(True)
        y = rowmlp.dropout_rows(drop, x)
        kept = y != 0
        assert abs(float(kept.float().mean()) - 0.5) < 0.01
        assert torch.equal(y[kept].float(), (x.detach()[kept].float() * 2).to(m.dtype).float())
        # columns / rows are not correlated with the mask in any obvious way
        assert float(kept.float().mean(0).std()) < 0.02 and float(kept.float().mean(1).std()) < 0.08
        g = torch.ones_like(y)
        y.backward(g)
        assert torch.equal(x.grad != 0, kept) and torch.equal(x.grad[kept].float(), torch.full_like(x.grad[kept], 2.0).float())
        y2 = rowmlp.dropout_rows(drop, x)
        assert not torch.equal(y2 != 0, kept)
        drop.p = 0.25
        assert abs(float((rowmlp.dropout_rows(drop, x) != 0).float().mean()) - 0.75) < 0.01
        drop.eval()
        assert rowmlp.dropout_rows(drop, x) is x


# ---- feature-space kNN through the split-bf16 screening pass (csrc/knn.hip: knn_screen_kernel) ---------------------------
def _knn_cloud(kind, B, N, D, g):
    x = torch.randn(B, N, D, generator=g)
    if kind == "clustered":            # tight clusters + exact duplicates: ties at the k-th distance
        c = torch.randn(B, 16, D, generator=g) * 3
        lab = torch.randint(0, 16, (B, N), generator=g)
        x = torch.gather(c, 1, lab.unsqueeze(-1).expand(B, N, D)) + 0.05 * x
        d = min(64, N // 2)
        x[:, N // 2:N // 2 + d] = x[:, :d]
    elif kind == "lattice":            # small integers: most distances tie exactly
        x = torch.randint(-2, 3, (B, N, D), generator=g).float()
    elif kind == "offset":             # |x|^2 >> neighbour distances: nothing certifies, everything is recomputed
        x = x * 0.01 + 50.0
    elif kind == "relu":               # what the network feeds it: BatchNorm + LeakyReLU rows
        x = torch.nn.functional.leaky_relu(x, 0.2)
    return x.cuda().contiguous()


@pytest.mark.parametrize("kind", ["gaussian", "clustered", "lattice", "offset", "relu"])
@pytest.mark.parametrize("N,D,k", [(1000, 64, 20), (2048, 128, 20), (777, 40, 8), (4096, 64, 20), (300, 32, 5),
                                   (1024, 96, 16), (20000, 32, 1), (64, 64, 20)])
def test_screened_knn_is_the_exact_knn(kind, N, D, k):
    """pcb_knn_screened against pcb_knn (itself bit-identical to the reference's topk on the fixtures,
    tests/test_gpu_ops.py): identical index lists on random, tied, duplicated and badly conditioned clouds."""
    from pointcloud_bridge_amd import ops
    g = torch.Generator().manual_seed(N + D + k)
    x = _knn_cloud(kind, 2, N, D, g)
    old = ops.set_screen_knn(False)
    try:
        exact = ops.knn(x, k)
        ops.set_screen_knn(True)
        ops.collect_knn_stats(True)
        screened = ops.knn(x, k)
        stats = ops.collect_knn_stats(False)
    finally:
        ops.set_screen_knn(old)
    assert len(stats) == 1, "the screening pass did not run"
    recomputed = int(stats[0][4].sum())
    print(kind, N, D, k, "recomputed", recomputed, "of", 2 * N)
    assert torch.equal(screened, exact)
    if kind in ("gaussian", "relu"):
        assert recomputed < 0.05 * 2 * N     # well separated neighbours certify
    if kind == "offset":
        assert recomputed > 0.5 * 2 * N      # (and the recomputation path is exercised)


def test_screened_knn_small_cloud_against_the_oracle():
    from oracle import oracle as orc
    from pointcloud_bridge_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.nn.functional.leaky_relu(torch.randn(2, 640, 64, generator=g), 0.2)
    assert np.array_equal(ops.knn(x.cuda(), 20).cpu().numpy(), orc.knn(x.numpy(), 20))


def test_two_models_keep_their_own_prefetches():
    """the prefetch tables belong to the model.  Model A prefetches its next batch, model B prefetches and
    runs in between (which used to clear A's pending pyramid: A then sampled inline and consumed the CPU generator a second
    time), then A runs: bit-identical to A alone, same number of CPU draws, both models' tables empty afterwards."""
    from pointcloud_bridge_amd.models import pointnet2_utils as pu
    a, b = _model(1).eval(), _model(2).eval()
    xa, ca, _ = _batch(11)
    xb, cb, _ = _batch(12)
    with torch.no_grad():
        torch.manual_seed(5)
        a.prefetch(xa)
        alone = a(xa, ca)
        after_alone = torch.rand(1)
        torch.manual_seed(5)
        a.prefetch(xa)
        state = torch.get_rng_state()          # B's own draws must not count against A's stream in this comparison
        b.prefetch(xb)
        b(xb, cb)
        torch.set_rng_state(state)
        mixed = a(xa, ca)
        after_mixed = torch.rand(1)
    assert torch.equal(alone, mixed)
    assert torch.equal(after_alone, after_mixed)
    assert a.sampling is not b.sampling
    assert not a.sampling.prefetched and not b.sampling.prefetched and not pu._default_state.prefetched


def test_scene_shard_is_a_property_of_the_model_or_an_argument():
    from pointcloud_bridge_amd.models import pointnet2_utils as pu
    xyz, _, _ = _batch(4, B=2, N=512)
    torch.manual_seed(3)
    whole = pu.farthest_point_sample(torch.cat([xyz, xyz]), 64)           # a global batch of 4 scenes
    torch.manual_seed(3)
    second = pu.farthest_point_sample(xyz, 64, shard=(1, 2))              # its second half on rank 1 of 2: same draws
    assert torch.equal(second, whole[2:])
    m = _model(1)
    m.sampling.scene_shard = (1, 2)
    with pu.sampling_scope(m.sampling):
        torch.manual_seed(3)
        assert torch.equal(pu.farthest_point_sample(xyz, 64), whole[2:])
    assert pu.scene_shard() == (0, 1)                                     # the process default is untouched
