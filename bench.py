#!/usr/bin/env python
"""Headline benchmark: points/sec, forward + backward (+ gradient all-reduce + Adam step) of the
PointNet++ segmentation network on synthetic clouds, B=16 scenes x N=16384 points per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus 8 --steps 20 --warmup 5

One process per GPU; scenes are sharded (weak scaling: B scenes per GPU; --scaling strong: ONE global
batch of B scenes split over the ranks); the only data-path collective is the flat gradient
all-reduce over RCCL, started bucket by bucket while the backward pass is still running.  Rank 0
prints ONE JSON line (contract in the task brief).
`roofline` is measured live with HIP events around the launches of the dominant HBM-bound kernel
inside the timed region; `roofline.whole_step` is the algorithmic traffic of EVERY launch of the
library over the step time, with the FPS chain's time as a separate latency term; `cpu_baseline`
times the ATen port of the reference's CPU path (oracle/torch_port.py) on this host's cores over a
bounded sample (rank 0, N=1 only); `extra` (N=1 only) carries short runs of the other configurations.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def _normalise(p):
    """Centre and scale to the unit ball exactly like utils/simpdataset.py:47-62."""
    p = p - p.mean(dim=1, keepdim=True)
    return p / p.norm(dim=-1).max(dim=1)[0].view(-1, 1, 1)


def synthetic_batch(B, N, seed, device, family="ball"):
    """family "ball": uniform in the unit ball.  family "bridge": the shape the authors train on --
    points on a deck plane, pier boxes and two cable/railing lines with strongly non-uniform density
    (ball-query early exit and FPS pruning depend on it), 80 % distinct points padded to N by
    repeating points like utils/simpdataset.py:143-149.  Colours U[0,1), labels 0..4."""
    g = torch.Generator().manual_seed(seed)
    if family == "ball":
        v = torch.randn(B, N, 3, generator=g)
        p = v / v.norm(dim=-1, keepdim=True) * torch.rand(B, N, 1, generator=g) ** (1.0 / 3.0)
    elif family == "bridge":
        M = int(N * 0.8)
        u = torch.rand(B, M, 3, generator=g)
        part = torch.rand(B, M, generator=g)
        deck = torch.stack([u[..., 0] * 40 - 20, u[..., 1] * 6 - 3, u[..., 2] * 0.05], -1)            # 55 %: a 40 x 6 m slab
        pier_x = (torch.randint(0, 4, (B, M), generator=g).float() - 1.5) * 10.0
        pier = torch.stack([pier_x + u[..., 0] * 1.0, u[..., 1] * 4 - 2, -u[..., 2] * 8.0], -1)      # 25 %: four piers
        rail_y = torch.where(u[..., 1] > 0.5, 3.0, -3.0)
        rail = torch.stack([u[..., 0] * 40 - 20, rail_y + u[..., 1] * 0.02, 1.0 + u[..., 2] * 0.1], -1)  # 12 %: railings
        ground = torch.stack([u[..., 0] * 50 - 25, u[..., 1] * 20 - 10, -8.0 - u[..., 2] * 0.5], -1)  # 8 %: ground patch
        p = torch.where((part < 0.55).unsqueeze(-1), deck,
                        torch.where((part < 0.80).unsqueeze(-1), pier, torch.where((part < 0.92).unsqueeze(-1), rail, ground)))
        pad = torch.randint(0, M, (B, N - M), generator=g)                                            # simpdataset.py:146-148
        p = torch.cat([p, torch.gather(p, 1, pad.unsqueeze(-1).expand(-1, -1, 3))], dim=1)
        perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
        p = torch.gather(p, 1, perm.unsqueeze(-1).expand(-1, -1, 3))
    else:
        raise ValueError(family)
    colors = torch.rand(B, N, 3, generator=g)
    labels = torch.randint(0, 5, (B, N), generator=g)
    return _normalise(p).contiguous().to(device), colors.to(device), labels.to(device)


def build_model(name, num_classes=5):
    from pointcloud_bridge_amd.models.containers import EnhancedPointNet2, PointNet2, PointNet2MSG
    from pointcloud_bridge_amd.models.DGCNN import DGCNN
    if name == "pn2_msg":
        return PointNet2MSG(num_classes), 1
    if name == "pn2_ssg":
        return PointNet2(num_classes, rgb_skip=True), 1
    if name == "dgcnn":
        return DGCNN(num_classes, k=20), 2
    if name == "bridgeseg":
        # the reference's whole BridgeSeg network (models/model.py:58-147; train_MulSca_BriStruNet_CB.py);
        # geometric1 and cls_head are constructed there and never called -- no gradient, no update
        net = EnhancedPointNet2(num_classes)
        for unused in (net.geometric1, net.cls_head):
            for p in unused.parameters():
                p.requires_grad_(False)
        return net, 1
    if name == "ptv3":
        # cfg5: inference_ptv3.py:101-105 (embed 384, depth 8, 2 heads -> head_dim 192), global attention
        from pointcloud_bridge_amd.models.PointTransformerV3 import PointTransformerV3
        return PointTransformerV3(num_classes, d_in=6, embed_dim=384, depth=8, num_heads=2), 2
    raise ValueError(name)


def loss_fn(logits, labels, channel_dim):
    """nn.CrossEntropyLoss() of the trainers (train_MulSca_PN2.py:161 on [B,C,N]; train_DGCNN.py:177-197 on
    [B*N,C]): GPU logits go through the library's one-pass kernel (losses.cross_entropy), the CPU port's through
    F.cross_entropy."""
    if logits.is_cuda:
        from pointcloud_bridge_amd.losses import cross_entropy
        return cross_entropy(logits, labels, channels_last=(channel_dim != 1))
    if channel_dim == 1:
        return F.cross_entropy(logits, labels)
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1))


def cpu_baseline(model_name, N, budget_s=25.0):
    """Reference CPU path (ATen port) on the host cores: fwd + CE + bwd, B=1 scenes of N points."""
    from oracle import torch_port as port
    # threads = the cores this process may use, capped at the 16-core share a one-GPU box grants
    # (os.cpu_count() reports the whole host and oversubscribing OpenMP stalls for minutes)
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("PCB_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: {model_name} B=1 N={N} on {cores} threads ...", file=sys.stderr, flush=True)
    torch.manual_seed(42)
    model, cdim = build_model(model_name)
    model.train()
    xyz, colors, labels = synthetic_batch(1, N, 0, "cpu")

    def step():
        model.zero_grad(set_to_none=True)
        loss_fn(port.run(model, xyz, colors), labels, cdim).backward()

    t0 = time.perf_counter()
    step()  # warm-up, also tells how many timed steps fit the budget
    first = time.perf_counter() - t0
    # a bounded sample of 10-20 s of CPU work: as many steps as fit `budget_s` after the warm-up, 20 at most
    n = max(1, min(20, int((min(budget_s, 16.0) - first) / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n
    return {"value": N / dt, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"{model_name} fwd+bwd, B=1 x N={N}, fp32, {n} timed steps after 1 warm-up "
                      f"({dt:.2f} s/step), oracle/torch_port.py (ATen port of the reference path)"}


class Run:
    """One timed configuration: model, data, optimiser, the step function, the timed loop."""

    RESIDENT_BATCHES = 4  # the loop walks over a few resident batches (step i prefetches the sampling of batch i+1)

    def __init__(self, args, model_name, precision, B, N, rank, world, device, mode="train", family="ball",
                 graph=False, loss="ce", sync_bn=False, strong=False):
        from pointcloud_bridge_amd import parallel, rowmlp
        from pointcloud_bridge_amd.models import pointnet2_utils as pu
        self.args, self.model_name, self.precision, self.mode, self.family = args, model_name, precision, mode, family
        self.B, self.N, self.rank, self.world, self.device = B, N, rank, world, device
        self.rowmlp = rowmlp
        self.coll = parallel.collectives()   # world > 1, or the one-rank RCCL rehearsal (PCB_DIST_SINGLE=1)
        rowmlp.set_precision(precision)
        torch.manual_seed(42)  # identical init on every rank; broadcast below makes it certain
        model, self.cdim = build_model(model_name)
        model = model.to(device).train()
        if sync_bn and world > 1:
            model = parallel.sync_batchnorm(model)
        parallel.broadcast_parameters(model)
        if args.no_dropout:  # equivalence runs: a sharded run cannot reproduce the single process's dropout masks
            for m in model.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
        self.model = model
        rowmlp.attach_step_operands(model)   # this run's own operand set: a captured step bakes in ITS tables and buffers only
        self.use_graph = mode == "train" and graph
        # inference through one captured hipGraph per batch (pn2_msg / pn2_ssg: the eager eval pass is bound by the
        # host -- ~160 launches, 2.4-2.8 ms of enqueueing against 2.6 ms of GPU time)
        self.infer_graph = mode == "infer" and graph and hasattr(model, "static_sampling") and model_name in ("pn2_msg", "pn2_ssg")
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        # train_MulSca_PN2.py:125: Adam(lr=1e-3, weight_decay=1e-4) -- as one fused update over a flat
        # parameter buffer (parallel.FlatAdam: torch's fused-Adam arithmetic, one launch instead of ~25)
        self.opt = parallel.FlatAdam(params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
        if self.use_graph:
            self.bucket = parallel.FlatGradAllReduce(params, keep_grad_tensors=True, assign_views=False)
        else:
            self.bucket = parallel.OverlappedGradAllReduce.by_children(model)
        # data: weak scaling = every rank its own batches; strong = one global batch, this rank's scenes
        self.batches = []
        for i in range(self.RESIDENT_BATCHES):
            if strong:
                x, c, lab = synthetic_batch(B * world, N, 1000 + 17 * i, "cpu", family)
                if getattr(args, "dup_halves", False):
                    h = B * world // 2
                    x, c, lab = (torch.cat([t[:h], t[:h]]) for t in (x, c, lab))
                sl = slice(rank * B, (rank + 1) * B)
                self.batches.append(tuple(t[sl].contiguous().to(device) for t in (x, c, lab)))
            else:
                self.batches.append(synthetic_batch(B, N, 1000 + 17 * i + 1000 * rank, device, family))
        pu.set_scene_shard(rank if strong else 0, world if strong else 1)
        torch.manual_seed(7 + (0 if strong else rank))  # CPU generator: FPS start indices
        self.prefetch = (not args.no_prefetch) and hasattr(model, "prefetch") and not self.use_graph
        if loss == "bridge":
            if self.cdim != 1:
                raise SystemExit("--loss bridge expects [B,C,N] logits (bridgeseg, pn2_msg, pn2_ssg)")
            from pointcloud_bridge_amd.losses import BridgeStructureLoss
            crit = BridgeStructureLoss(alpha=80, rel_margin=0.3).to(device)  # train_MulSca_BriStruNet_CB.py:151-156
            self.loss_of = lambda logits, batch: crit(logits.float(), batch[2], batch[0])
        else:
            self.loss_of = lambda logits, batch: loss_fn(logits, batch[2], self.cdim)
        self.i = 0
        if mode == "infer":
            model.eval()
        if self.use_graph:
            self._capture()
        if self.infer_graph:
            self._capture_infer()

    # -- steps ------------------------------------------------------------------------------------
    def _batch(self, k=0):
        return self.batches[(self.i + k) % len(self.batches)]

    def train_step(self):
        batch = self._batch()
        self.bucket.zero()
        loss = self.loss_of(self.model(batch[0], batch[1]), batch)
        if self.prefetch:
            self.model.prefetch(self._batch(1)[0])  # sampling pyramid of the NEXT batch, concurrent with this backward
        loss.backward()
        flat = self.bucket.finish()
        if self.args.dump and self.i == 0:
            self.first_grad = flat.detach().clone()  # the averaged gradient of the first step (equivalence tests)
        self.opt.step(flat)
        self.rowmlp.prepare_step(self.model)   # the operands of every stack from the new weights: one launch (rowmlp.prepare_step)
        self.i += 1
        return loss

    def infer_step(self):
        if self.infer_graph:
            return self.graph_infer_step()
        batch = self._batch()
        with torch.no_grad():
            if self.prefetch and hasattr(self.model, "set_next"):
                self.model.set_next(self._batch(1)[0])  # pipelined serving: the next batch's FPS runs beside this pass's decoder
            loss = self.loss_of(self.model(batch[0], batch[1]), batch)
        self.i += 1
        return loss

    def _capture_infer(self):
        """The eval-mode forward pass (+ loss) of one batch as ONE hipGraph; the sampling pyramid, ball queries and
        decoder k-NN of the NEXT batch run inside it on a forked stream, into the staging set the next replay
        commits.  FPS start indices are drawn on the host before every replay (StaticSampling.draw), as in the
        captured training step."""
        from pointcloud_bridge_amd.models import pointnet2_utils as pu
        model = self.model
        xyz, colors, labels = (t.clone() for t in self.batches[0])
        xyz_next = xyz.clone()
        batch = (xyz, colors, labels)
        static = model.static_sampling(xyz)
        pu.set_static_sampling(static)
        loss_buf = torch.zeros((), device=self.device)

        def fwd():
            with torch.no_grad():
                static.commit()
                # forked at the TOP of the pass: its chain (three FPS levels, ball queries, k-NN: ~2 ms on one stream) is
                # longer than the decoder, and a captured pass must join it before it ends -- forked behind the encoder
                # (what the eager pipeline does via model.set_next) it was the critical path: 3.16 ms against 2.61
                static.compute_beside(xyz_next, calls=12)
                self.logits_out = model(xyz, colors)     # (captured: the replay's output lives in the graph's pool)
                loss = self.loss_of(self.logits_out, batch)
                static.join()
                loss_buf.copy_(loss)

        with torch.no_grad():
            static.draw()
            static.compute(xyz_next)              # what the first pass commits
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(3):                    # eager warm-up on the capture stream (fills the eval operand cache)
                static.draw()
                fwd()
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        static.draw()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            fwd()

        def graph_infer_step():
            cur, nxt = self._batch(), self._batch(1)
            xyz.copy_(cur[0]); colors.copy_(cur[1]); labels.copy_(cur[2])
            xyz_next.copy_(nxt[0])
            self.i += 1
            static.draw()
            graph.replay()
            return loss_buf

        self.graph_infer_step, self.static = graph_infer_step, static

    def _capture(self):
        """Forward + loss + backward (+ the next step's sampling pyramid on a forked stream) replayed as
        ONE hipGraph.  The CPU-generator draws for FPS stay on the host (StaticSampling.draw), the
        gradient all-reduce and the fused Adam step stay outside the graph.  One resident batch (the
        graph's tensors are fixed)."""
        from pointcloud_bridge_amd import ops
        from pointcloud_bridge_amd.models import pointnet2_utils as pu
        model, bucket, opt = self.model, self.bucket, self.opt
        # the graph's tensors are fixed: the step's batch is copied into them before every replay, and the NEXT
        # batch's coordinates into `xyz_next`, from which the captured step computes the next sampling pyramid
        xyz, colors, labels = (t.clone() for t in self.batches[0])
        xyz_next = xyz.clone()
        batch = (xyz, colors, labels)
        static = None
        if hasattr(model, "static_sampling"):
            static = model.static_sampling(xyz)
            pu.set_static_sampling(static)
            static.draw()
            static.compute(xyz_next)
        side = torch.cuda.Stream()
        loss_buf = torch.zeros((), device=self.device)
        want = getattr(self.args, "graph_segments", 0) or (2 if self.coll else 1)
        if want == 2 and hasattr(model, "_cut"):
            return self._capture_two_segments(model, opt, xyz, colors, labels, xyz_next, batch, static, side, loss_buf)

        def fwd_bwd():
            self.rowmlp.prepare_step(model)      # operands of every stack from the weights the last Adam step left
            if static is not None:
                static.commit()                  # what the previous step computed for this batch becomes live
            loss = self.loss_of(model(xyz, colors), batch)
            if static is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    static.compute(xyz_next)     # pyramid of the next step's batch, beside the backward pass
                # the persistent backward GEMMs of the first stacks leave the pyramid's CUs alone (ops.apply_concurrency_hint)
                ops.set_background_work(torch.cuda.Event(), 2 * xyz.shape[0])
            loss.backward()
            if static is not None:
                torch.cuda.current_stream().wait_stream(side)
            loss_buf.copy_(loss.detach())

        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(3):                   # eager warm-up on the capture stream
                bucket.zero()
                if static is not None:
                    static.draw()
                fwd_bwd()
                bucket.reduce()
                opt.step(bucket.flat)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        bucket.zero()
        graph = torch.cuda.CUDAGraph()
        if static is not None:
            static.draw()
        # (thread_local: the process group's watchdog thread may query events while this thread captures)
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            fwd_bwd()
        grads = [p.grad for p in self.params if p.grad is not None]  # the graph's fixed tensors

        def load_batches():
            cur, nxt = self._batch(), self._batch(1)
            xyz.copy_(cur[0]); colors.copy_(cur[1]); labels.copy_(cur[2])
            xyz_next.copy_(nxt[0])
            self.i += 1

        def graph_step():
            load_batches()
            if static is not None:
                static.draw()
            graph.replay()
            bucket.reduce()
            opt.step(bucket.flat)
            return loss_buf

        def eager_step():
            """The same step launched kernel by kernel (HIP events around the roofline kernel need
            real launches); gradients accumulate into the graph's tensors, zeroed first."""
            load_batches()
            if static is not None:
                static.draw()
            torch._foreach_zero_(grads)
            fwd_bwd()
            bucket.reduce()
            opt.step(bucket.flat)
            return loss_buf

        self.graph_step, self.eager_step, self.static = graph_step, eager_step, static

    def _capture_two_segments(self, model, opt, xyz, colors, labels, xyz_next, batch, static, side, loss_buf):
        """The captured step as TWO hipGraphs split at the seam between encoder and decoder, so that the gradient exchange
        overlaps the backward pass in captured mode too (one flat all-reduce behind the whole replay left
        the links idle during the backward pass and the chip idle during the exchange):
            graph A   forward, loss, backward of head + decoder down to the seam; their gradients packed into the tail
                      of the flat gradient buffer
            (host)    asynchronous all-reduce of that tail -- RCCL's stream; it waits for graph A only
            graph B   backward of the encoder from the seam's gradients; packed into the head of the flat buffer
            (host)    all-reduce of the head, wait for both, average, fused Adam
        The seam: models.containers._SamplingPrefetchMixin.decoder_cut detaches what the encoder hands to the decoder;
        autograd runs from the loss to the detached tensors (A) and from the encoder's outputs with those gradients (B).
        Both graphs share one memory pool (B reads what A's forward pass saved)."""
        from pointcloud_bridge_amd import ops
        names = [n for n, _ in model.named_children()]
        k = names.index(model.decoder_first)
        enc_ids = {id(p) for _, child in list(model.named_children())[:k] for p in child.parameters()}
        E = [p for p in self.params if id(p) in enc_ids]
        D = [p for p in self.params if id(p) not in enc_ids]
        if [id(p) for p in E + D] != [id(p) for p in self.params]:
            raise SystemExit("--graph-segments 2: the encoder's parameters are not a prefix of the parameter list")
        nE = sum(p.numel() for p in E)
        flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=self.device)
        self.bucket.flat = flat
        seam = {}

        def cut(*ts):
            seam["enc"] = ts
            seam["dec"] = tuple(t.detach().requires_grad_(t.requires_grad) for t in ts)
            return seam["dec"]

        model.decoder_cut = cut

        def pack(grads, params, out):
            torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(grads, params)], out=out)

        def seg_a():
            self.rowmlp.prepare_step(model)
            if static is not None:
                static.commit()
            loss = self.loss_of(model(xyz, colors), batch)
            if static is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    static.compute(xyz_next)
                ops.set_background_work(torch.cuda.Event(), 2 * xyz.shape[0])
            dec = [t for t in seam["dec"] if t.requires_grad]
            grads = torch.autograd.grad(loss, dec + D, allow_unused=True)
            seam["bgrads"] = [g if g is not None else torch.zeros_like(t) for g, t in zip(grads[:len(dec)], dec)]
            pack(grads[len(dec):], D, flat[nE:])
            if static is not None:
                torch.cuda.current_stream().wait_stream(side)
            loss_buf.copy_(loss.detach())

        def seg_b():
            enc = [t for t in seam["enc"] if t.requires_grad]
            pack(torch.autograd.grad(enc, E, grad_outputs=seam["bgrads"], allow_unused=True), E, flat[:nE])

        def exchange_tail():
            return dist.all_reduce(flat[nE:], op=dist.ReduceOp.SUM, async_op=True) if self.coll else None

        def exchange_head(work):
            if self.coll:
                dist.all_reduce(flat[:nE], op=dist.ReduceOp.SUM)
                work.wait()
                flat.div_(self.world)

        self.segment_log = []   # (tests) the order in which segments and exchanges were issued in the last step

        def run_step(a, b):
            self.segment_log = ["A"]
            a()
            work = exchange_tail()
            self.segment_log += ["allreduce(decoder)", "B"]
            b()
            exchange_head(work)
            self.segment_log += ["allreduce(encoder)"]
            opt.step(flat)

        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(3):                   # eager warm-up on the capture stream
                if static is not None:
                    static.draw()
                run_step(seg_a, seg_b)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        if static is not None:
            static.draw()
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            seg_a()
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
            seg_b()

        def load_batches():
            cur, nxt = self._batch(), self._batch(1)
            xyz.copy_(cur[0]); colors.copy_(cur[1]); labels.copy_(cur[2])
            xyz_next.copy_(nxt[0])
            self.i += 1

        def graph_step():
            load_batches()
            if static is not None:
                static.draw()
            run_step(ga.replay, gb.replay)
            return loss_buf

        def eager_step():
            load_batches()
            if static is not None:
                static.draw()
            run_step(seg_a, seg_b)
            return loss_buf

        self.graph_step, self.eager_step, self.static, self.segments = graph_step, eager_step, static, 2

    # -- the timed loop ---------------------------------------------------------------------------
    def fence(self):
        torch.cuda.synchronize()
        if self.coll:
            dist.barrier()
        torch.cuda.synchronize()

    def isolated_launches(self, steps=2, warmup=3):
        """HIP-event durations of the roofline kernel family with every chain of the model on ONE stream.
        In the timed region the independent chains of a module run on branch streams
        (pointnet2_utils.run_branches) and two GEMMs that share the chip each take longer than alone: a
        launch's own duration -- what a roofline fraction is about -- needs steps without that.  Run AFTER the
        timed region and not inside it: switching moves the step's tensors to another stream's allocator pool
        (a burst of hipMalloc calls), which would cost the timed steps around it more than a millisecond each."""
        from pointcloud_bridge_amd import ops
        from pointcloud_bridge_amd.models import pointnet2_utils as pu
        if self.mode != "train" or not (self.use_graph or pu.branch_streams_enabled()):
            return None
        step = self.eager_step if self.use_graph else self.train_step
        branches = pu.branch_streams_enabled()
        pu.set_branch_streams(False)
        try:
            for _ in range(warmup):
                step()
            self.fence()
            ops.kernel_timer_start()
            for _ in range(steps):
                ops.kernel_timer_enable(True)
                step()
            self.fence()
            launches, kernel_ms, nt_bytes = ops.kernel_timer_stop()
            all_n, _, all_bytes = ops.kernel_timer_read(-1)
        finally:
            pu.set_branch_streams(branches)
        return {"nt_launches": launches, "nt_ms": kernel_ms, "nt_bytes": nt_bytes,
                "lib_launches_per_step": all_n / steps, "lib_bytes_per_step": all_bytes / steps}

    def probe(self, steps=5, warmup=5, windows=3):
        """Wall time per step over a few steps (choice of the execution mode, see choose_exec): the fastest of
        `windows` windows of `steps` steps -- single windows were seen 50 % off on some boxes (13.2 ms for a mode
        whose timed region then ran at 8.1), which flipped the choice."""
        step = (self.graph_step if self.use_graph else self.train_step) if self.mode == "train" else self.infer_step
        for _ in range(warmup):
            step()
        gc.collect()   # measured like the timed region: no cyclic collections inside (see timed)
        gc.disable()
        dt = float("inf")
        for _ in range(windows):
            self.fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            self.fence()
            dt = min(dt, (time.perf_counter() - t0) / steps)
        gc.enable()
        if self.coll:
            t = torch.tensor([dt], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed(self, steps, warmup, roofline=True):
        from pointcloud_bridge_amd import ops
        step = (self.graph_step if self.use_graph else self.train_step) if self.mode == "train" else self.infer_step
        losses = []
        for _ in range(warmup):
            step()
        # The interpreter's cyclic garbage collector stops the host for tens of milliseconds at a time
        # (a step creates thousands of short-lived tensor objects); such a pause drains the GPU queue.
        # As in production training loops it is run at chosen points instead: here, before and after
        # the timed steps.
        gc.collect()
        gc.disable()
        self.fence()
        ops.kernel_timer_start()
        t0 = time.perf_counter()
        host_s = 0.0  # time the host spends enqueueing (diagnostic: host-bound vs GPU-bound)
        nsampled = 0
        for i in range(steps):
            # HIP events around the roofline kernel on every 10th step (launched eagerly in graph mode)
            # (a captured step launches nothing through the library: its launch durations come from eager steps
            # outside the timed region, Run.isolated_launches)
            sampled = bool(roofline and i % 10 == 0 and not self.use_graph and not os.environ.get("PCB_BENCH_NO_ROOFLINE"))
            nsampled += sampled
            ops.kernel_timer_enable(sampled)
            h0 = time.perf_counter()
            loss = (self.eager_step if (self.use_graph and sampled) else step)()
            host_s += time.perf_counter() - h0
            if self.args.dump:
                lg = loss.detach().clone()
                if self.coll:  # the mean over ALL scenes, as a single process would report it
                    dist.all_reduce(lg, op=dist.ReduceOp.SUM)
                    lg /= self.world
                losses.append(lg)
        self.fence()
        dt = time.perf_counter() - t0
        gc.enable()
        launches, kernel_ms, nt_bytes = ops.kernel_timer_stop()
        fps_n, fps_ms, _ = ops.kernel_timer_read(1)
        all_n, _, all_bytes = ops.kernel_timer_read(-1)
        if self.coll:
            t = torch.tensor([dt], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # (a graph replay launches nothing through the library's entry points: its bytes are those of the eager steps)
        counted_steps = nsampled if self.use_graph else steps
        res = {"ms_per_step": dt / steps * 1e3, "points_per_s": self.world * self.B * self.N / (dt / steps),
               "loss": float(loss.detach()), "host_enqueue_ms_per_step": host_s / steps * 1e3,
               "nt_launches": launches, "nt_ms": kernel_ms, "nt_bytes": nt_bytes,
               "fps_ms_per_step": fps_ms / nsampled if (fps_n and nsampled) else None,
               "lib_launches_per_step": all_n / max(counted_steps, 1),
               "lib_bytes_per_step": all_bytes / max(counted_steps, 1)}
        if self.args.dump:
            res["losses"] = [float(x) for x in losses]
        return res

    def close(self):
        from pointcloud_bridge_amd.models import pointnet2_utils as pu
        if self.use_graph or self.infer_graph:
            pu.set_static_sampling(None)
        pu.set_scene_shard(0, 1)
        if hasattr(self.bucket, "close"):
            self.bucket.close()
        if getattr(self.model, "decoder_cut", None) is not None:
            self.model.decoder_cut = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="pn2_msg", choices=["pn2_msg", "pn2_ssg", "dgcnn", "bridgeseg", "ptv3"])
    ap.add_argument("--batch", type=int, default=None,
                    help="scenes per GPU (default 16; dgcnn 8); with --scaling strong: scenes of the GLOBAL batch")
    ap.add_argument("--npoints", type=int, default=None, help="points per scene (default 16384; dgcnn 8192)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: B scenes per GPU (the default); strong: ONE global batch of B scenes split over the "
                         "ranks, FPS start indices drawn for the global batch on every rank (SURVEY 8e)")
    ap.add_argument("--data", default="ball", choices=["ball", "bridge"],
                    help="synthetic family: unit-ball clouds, or bridge-like clouds (deck / piers / railings, "
                         "non-uniform density, 20 %% repeated points as the reference's padding produces)")
    ap.add_argument("--exec", default="auto", choices=["auto", "eager", "graph"],
                    help="how the training step is issued: eager (kernel by kernel, branch streams), graph (one captured "
                         "hipGraph per step), auto (time a few steps of both, keep the faster)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short runs of the other configurations")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not overlap the next batch's FPS pyramid with the current backward pass")
    ap.add_argument("--sync-bn", action="store_true",
                    help="SyncBatchNorm across ranks (both precisions: statistics all-reduced inside the fused engine)")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train: fwd+loss+bwd+all-reduce+Adam (the headline metric); infer: eval-mode forward only, "
                         "the reference's own published metric (eva_model.py:137-168, model_performance_comparison.csv)")
    ap.add_argument("--graph", action="store_true",
                    help="train mode: replay forward+backward as one captured hipGraph instead of launching every "
                         "kernel from the host (pn2_msg / pn2_ssg / dgcnn)")
    ap.add_argument("--loss", default="ce", choices=["ce", "bridge"],
                    help="bridge: BridgeStructureLoss(alpha=80, rel_margin=0.3) of train_MulSca_BriStruNet_CB.py:151-156 "
                         "(bridgeseg / pn2_msg / pn2_ssg logits [B,C,N])")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="row type of the fused MLP engine: bf16 activations (BASELINE config 2) or fp32 rows (parity mode)")
    ap.add_argument("--dump", default=None, help="rank 0 writes step losses and the final parameters to this .pt file")
    ap.add_argument("--no-dropout", action="store_true", help="Dropout layers with p = 0 (sharded-vs-single equivalence runs)")
    ap.add_argument("--graph-segments", type=int, default=0, choices=[0, 1, 2],
                    help="captured step as ONE hipGraph (1) or as TWO split at the encoder/decoder seam, the decoder's gradient "
                         "bucket all-reduced while the encoder's backward segment replays (2); 0 = 2 with more than one rank, else 1")
    ap.add_argument("--dup-halves", action="store_true",
                    help="--scaling strong: the second half of the global batch repeats the first (equivalence runs of a criterion "
                         "with batch-level statistics -- BridgeStructureLoss: every rank's shard then has the global batch's)")
    args = ap.parse_args()

    from pointcloud_bridge_amd import parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP kernels, no CPU fallback)")
    local = local % torch.cuda.device_count()  # ranks may share a GPU in a gloo rehearsal
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    B = args.batch or (8 if args.model in ("dgcnn", "ptv3") else 16)
    N = args.npoints or (8192 if args.model == "dgcnn" else 4096 if args.model == "ptv3" else 16384)  # ptv3: inference_ptv3.py:48-51
    strong = args.scaling == "strong"
    if strong:
        if B % world:
            raise SystemExit(f"--scaling strong: the global batch of {B} scenes does not split over {world} ranks")
        B //= world

    def make(graph):
        return Run(args, args.model, args.precision, B, N, rank, world, device, args.mode, args.data, graph, args.loss,
                   args.sync_bn, strong)

    # Execution mode of the training step.  The eager step (branch streams, sampling of the next batch on a side
    # stream) needs 7.7-10 ms of host time per step depending on the box's host cores -- as much as the GPU needs --
    # while the captured step (one hipGraph replay + all-reduce + Adam) needs almost none but keeps ball query and
    # k-NN of the step on its main path.  Which one is faster depends on the host: `auto` times a few steps of both
    # and runs the timed region with the faster one (the same decision on every rank: maximum over ranks).
    exec_mode = "graph" if args.graph else args.exec
    probes = {}
    can_graph = (args.mode == "train" and args.model in ("pn2_msg", "pn2_ssg", "dgcnn", "bridgeseg") and args.loss in ("ce", "bridge")
                 and not args.dump and not args.no_prefetch)
    if exec_mode == "auto" and args.mode == "infer" and args.model in ("pn2_msg", "pn2_ssg") and not args.no_prefetch:
        exec_mode = "graph"     # the eval pass is host-bound when launched kernel by kernel (2.77 vs 2.61 ms captured)
    if exec_mode == "auto" and not can_graph:
        exec_mode = "eager"
    run = None
    if exec_mode == "auto":
        eager = make(False)
        probes["eager_ms"] = eager.probe() * 1e3
        graph, ok = None, 1
        try:
            graph = make(True)
        except Exception as e:  # a capture that fails must not cost the run
            ok = 0
            probes["graph_error"] = f"{type(e).__name__}: {e}"[:200]
        if parallel.collectives():
            # the two modes exchange gradients differently (buckets vs one flat all-reduce): every rank must take
            # the same one, so a capture that failed anywhere sends all ranks to the eager step
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            probes["graph_ms"] = graph.probe() * 1e3
        elif graph is not None:
            graph.close()
            graph = None
        if graph is not None and probes["graph_ms"] < probes["eager_ms"]:
            exec_mode, run, other = "graph", graph, eager
        else:
            exec_mode, run, other = "eager", eager, graph
        # launch durations of the roofline family: from eager single-stream steps of whichever run is dropped (the
        # kernels are the same ones), before it goes
        from pointcloud_bridge_amd.models import pointnet2_utils as pu
        iso_other = None
        if other is not None and not os.environ.get("PCB_BENCH_NO_ROOFLINE"):
            if other is eager:
                pu.set_static_sampling(None)          # the eager step samples through its own prefetch pipeline
            iso_other = other.isolated_launches()
        if other is not None:
            other.close()
        other = eager = graph = None   # really released: nothing of the dropped run outlives this point (its operand set is its own)
        if run.use_graph:
            pu.set_static_sampling(run.static)        # (only the eager steps of the captured run read it)
        pu.set_scene_shard(rank if strong else 0, world if strong else 1)
        torch.cuda.empty_cache()
    else:
        iso_other = None
        run = make(exec_mode == "graph")
    res = run.timed(args.steps, args.warmup)
    # (every rank: the steps contain the gradient all-reduce)
    iso = None
    if not (os.environ.get("PCB_BENCH_NO_ROOFLINE") or args.dump):
        iso = iso_other if iso_other is not None else run.isolated_launches()
    if args.dump and rank == 0:
        torch.save({"losses": res["losses"], "flat": run.opt.flat.detach().cpu(),
                    "first_grad": getattr(run, "first_grad", torch.zeros(0)).cpu()}, args.dump)
    prefetching = run.prefetch
    run.close()

    if rank == 0:
        # BASELINE.md publishes one number on this path that a bench mode reproduces exactly:
        # PointNet2 (SSG) inference, B=4 x N=4096, fp32, eval, 1 GPU (RTX 4090): 35 557 pts/s
        # (Highway_bridge/model_performance_comparison.csv:4).  The headline fwd+bwd metric has none.
        vs_baseline = None
        if (args.mode == "infer" and args.model == "pn2_ssg" and B == 4 and N == 4096
                and args.precision == "fp32" and world == 1):
            vs_baseline = res["points_per_s"] / 35557.0
        traffic = None  # HBM bytes per launch from the committed PMC passes of this same workload
        for pmc_name in ("r03_pmc_gemm_nt_bf16.json", "r02_pmc_gemm_nt_bf16.json", "r01_pmc_gemm_nt_bf16.json"):
            pmc = os.path.join(REPO, "profiles", pmc_name)
            if (args.model == "pn2_msg" and args.precision == "bf16" and B == 16 and N == 16384 and args.data == "ball"
                    and args.mode == "train" and os.path.exists(pmc)):
                with open(pmc) as f:
                    traffic = json.load(f)["traffic_bytes_per_launch"]
                break
        concurrent = None
        if iso and iso["nt_launches"]:
            # the timed region's figures (GEMMs of concurrent chains share the chip) go along for the record
            if res["nt_launches"]:
                c_n = res["nt_launches"]
                concurrent = {"launches": c_n, "avg_launch_us": res["nt_ms"] / c_n * 1e3,
                              "achieved": (res["nt_bytes"] / c_n) / (res["nt_ms"] / c_n * 1e-3) / 1e9}
            res = dict(res, **iso)
        launches = res["nt_launches"]
        alg_bytes = res["nt_bytes"] / max(launches, 1)   # per launch
        avg_s = res["nt_ms"] / max(launches, 1) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if launches else 0.0
        step_s = res["ms_per_step"] * 1e-3
        whole = res["lib_bytes_per_step"] / step_s / 1e9
        workload = (f"{args.model} "
                    + (("fwd+" + ("CE" if args.loss == "ce" else "BridgeStructureLoss") + "+bwd+"
                        + ("two graph segments, the decoder's gradient bucket all-reduced beside the encoder's backward segment"
                           if (exec_mode == "graph" and getattr(run, "segments", 1) == 2) else
                           "flat gradient packing" if world == 1 else
                           ("one flat grad-allreduce" if exec_mode == "graph" else "overlapped bucketed grad-allreduce"))
                        + "+Adam" + (", captured hipGraph step" if exec_mode == "graph" else ""))
                       if args.mode == "train" else
                       ("eval-mode forward+CE" + (", next batch FPS pipelined" if prefetching else "")))
                    + f", B={B} scenes/GPU x N={N} pts, "
                    + ("unit-ball clouds" if args.data == "ball" else "bridge-like clouds (deck/piers/railings, 20% repeated points)")
                    + f", {Run.RESIDENT_BATCHES} resident batches in rotation (configs[1] of BASELINE.json)")
        out = {
            "metric": (("points/sec fwd+bwd, " if args.mode == "train" else "points/sec inference (eval forward), ")
                       + {"dgcnn": f"DGCNN k=20 EdgeConv seg N={N} B={B}",
                          "bridgeseg": f"BridgeSeg (EnhancedPointNet2: bridge encoders + PointNet++ MSG) N={N} B={B}"}
                       .get(args.model, f"PointTransformerV3 (global attention) N={N} B={B}" if args.model == "ptv3" else
                            f"PointNet++ seg N={N} B={B * world if strong else B}")),
            "value": res["points_per_s"],
            "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": vs_baseline, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": workload, "scenes_per_gpu": B, "points_per_scene": N,
                       "parallelism": f"dp{world} (scenes sharded, {args.scaling} scaling"
                                      + (", SyncBatchNorm" if args.sync_bn and world > 1 else "") + ")",
                       "parity": ("bf16 rows against the reference's fp32 fixtures: indices bit-identical, eval logits 5e-3, train-mode "
                                  "logits 1e-1 (the network's own noise amplification), loss 5e-4, gradient norms 2e-2 median; "
                                  "the 1e-4 logit bar is met by --precision fp32 on the same engine (DESIGN.md section 2)")
                                 if args.precision == "bf16" else
                                 "fp32 rows: logits within 1e-4 of the reference's fixtures, indices bit-identical (DESIGN.md section 2)",
                       "loss": res["loss"], "host_enqueue_ms_per_step": res["host_enqueue_ms_per_step"],
                       "library_launches_per_step": res["lib_launches_per_step"], "graph": exec_mode == "graph",
                       "exec": dict({"mode": exec_mode, "requested": "graph" if args.graph else args.exec}, **probes)},
            "roofline": {"bound": "hbm", "kernel": ROOFLINE_KERNEL, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launches": launches, "avg_launch_us": avg_s * 1e6,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "sampling": ("HIP events around every launch of the family in 2 eager steps outside the timed "
                                      "region with all chains of the model on one stream (a launch's own duration); "
                                      "`timed_region` (eager execution only) holds the same measurement on every 10th "
                                      "timed step, where GEMMs of concurrent chains share the chip") if iso else
                                     "HIP events around every launch of the family on every 10th timed step",
                         "timed_region": concurrent,
                         # algorithmic bytes of ALL kernel families of the library over the step time
                         # (ATen glue not counted), with the sampling chain's time as a separate latency term
                         "whole_step": {"algorithmic_bytes_per_step": res["lib_bytes_per_step"], "achieved": whole,
                                        "frac": whole / HBM_PEAK_GBS, "unit": "GB/s",
                                        "fps_ms_per_step": res["fps_ms_per_step"],
                                        "fps_note": "FPS pyramid of the next batch: a latency chain on a side stream "
                                                    "(1 workgroup per scene), concurrent with the backward pass"}},
        }
        if world == 1 and not args.no_extras and args.model == "pn2_msg" and args.mode == "train" and not args.dump:
            out["extra"] = extras(args, device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, N)
        print(json.dumps(out), flush=True)
    if parallel.collectives():
        dist.barrier()
        dist.destroy_process_group()


def knn_roofline(B, N, device, D=64, k=20):
    """cfg3's dominant kernel: the feature-space kNN call of an EdgeConv block (models/DGCNN.py:49-70 at D = 64, three per
    step).  Its pair products run on the bf16 matrix core (three MFMAs per 16 channels of the two-term split) and its
    selection on the vector pipe; priced against the dense bf16 MFMA peak: algorithmic work = the 2 N^2 D flop per scene of
    the distance matrix, duration = HIP events on the launch stream around 10 calls on BatchNorm+LeakyReLU-like rows."""
    from pointcloud_bridge_amd import ops
    torch.manual_seed(3)
    x = torch.nn.functional.leaky_relu(torch.randn(B, N, D, device=device), 0.2).contiguous()
    for _ in range(3):
        ops.knn(x, k)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        ops.knn(x, k)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 10 * 1e3
    flops = 2.0 * B * N * N * D
    peak = 2500.0   # dense bf16 MFMA, TFLOP/s (MI355X_MICROARCH.md)
    return {"bound": "mfma", "kernel": "pcb_knn_screened (knn_screen_kernel<64,20,24> + split / norms / exact fallback)",
            "achieved": flops / (us * 1e-6) / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flops / (us * 1e-6) / 1e12 / peak,
            "traffic": None, "avg_call_us": us,
            "note": "distance-matrix flop (2 N^2 D per scene) per call; the kernel issues 3x that on the bf16 matrix core and is "
                    "bound by the vector instructions of its top-k selection (DESIGN.md section 6), not by the products"}


def extras(args, device):
    """Short runs (a few steps each) of the other configurations SURVEY section 8(d) names, so that the
    driver's one bench line carries them: cfg3 DGCNN, cfg4's network BridgeSeg, the fp32 parity mode
    of the headline network, the headline network on bridge-like clouds, and the captured-graph step."""
    out = {}
    plan = [
        ("pn2_msg_bridge_like_B16_N16384_bf16", dict(model_name="pn2_msg", precision="bf16", B=16, N=16384, family="bridge")),
        # the same clouds through the captured step (what --exec auto runs where the host is the bound)
        ("pn2_msg_bridge_like_graph_B16_N16384_bf16", dict(model_name="pn2_msg", precision="bf16", B=16, N=16384, family="bridge", graph=True)),
        ("pn2_msg_graph_B16_N16384_bf16", dict(model_name="pn2_msg", precision="bf16", B=16, N=16384, graph=True)),
        ("pn2_msg_fp32_B16_N16384", dict(model_name="pn2_msg", precision="fp32", B=16, N=16384)),
        # the single-scale network the reference publishes its own numbers on (models/pointnet2.py), captured step
        ("pn2_ssg_graph_B16_N16384_bf16", dict(model_name="pn2_ssg", precision="bf16", B=16, N=16384, graph=True)),
        ("dgcnn_k20_B8_N8192_bf16", dict(model_name="dgcnn", precision="bf16", B=8, N=8192)),
        ("dgcnn_k20_graph_B8_N8192_bf16", dict(model_name="dgcnn", precision="bf16", B=8, N=8192, graph=True)),
        ("bridgeseg_B16_N16384_bf16", dict(model_name="bridgeseg", precision="bf16", B=16, N=16384)),
        # cfg4's network through the captured step (the eager step is bound by the host: ~770 launches), with the
        # cross-entropy of the other rows and with the criterion cfg4 trains with (train_MulSca_BriStruNet_CB.py:151-156)
        ("bridgeseg_graph_B16_N16384_bf16", dict(model_name="bridgeseg", precision="bf16", B=16, N=16384, graph=True)),
        ("bridgeseg_bridge_loss_graph_B16_N16384_bf16", dict(model_name="bridgeseg", precision="bf16", B=16, N=16384, graph=True,
                                                             loss="bridge")),
        ("pn2_msg_infer_B16_N16384_bf16", dict(model_name="pn2_msg", precision="bf16", B=16, N=16384, mode="infer")),
        ("pn2_msg_infer_graph_B16_N16384_bf16", dict(model_name="pn2_msg", precision="bf16", B=16, N=16384, mode="infer", graph=True)),
        # the ONE configuration the reference publishes a number for (model_performance_comparison.csv:4: PointNet2 SSG, eval
        # forward, B=4 x N=4096, fp32, 35 557 points/s on an RTX 4090): `vs_published` in this row
        ("pn2_ssg_infer_published_B4_N4096_fp32", dict(model_name="pn2_ssg", precision="fp32", B=4, N=4096, mode="infer", graph=True)),
        # cfg5 as the reference runs it (inference_ptv3.py:48-51, :101-105) and on tiles of 16384 points
        ("ptv3_infer_B8_N4096_bf16", dict(model_name="ptv3", precision="bf16", B=8, N=4096, mode="infer")),
        ("ptv3_infer_B2_N16384_bf16", dict(model_name="ptv3", precision="bf16", B=2, N=16384, mode="infer")),
    ]
    for name, kw in plan:
        try:
            torch.cuda.empty_cache()
            run = Run(args, rank=0, world=1, device=device, **kw)
            res = run.timed(40, 10) if kw.get("mode") == "infer" else run.timed(8, 4)   # (an inference pass is 1-3 ms)
            run.close()
            launches = res["nt_launches"]
            entry = {"ms_per_step": res["ms_per_step"], "points_per_s": res["points_per_s"], "loss": res["loss"],
                     "host_enqueue_ms_per_step": res["host_enqueue_ms_per_step"],
                     "library_launches_per_step": res["lib_launches_per_step"],
                     "whole_step_GBps": res["lib_bytes_per_step"] / (res["ms_per_step"] * 1e-3) / 1e9,
                     "fps_ms_per_step": res["fps_ms_per_step"]}
            if launches:
                entry["gemm_nt_GBps"] = res["nt_bytes"] / (res["nt_ms"] * 1e-3) / 1e9
                entry["gemm_nt_avg_us"] = res["nt_ms"] / launches * 1e3
                if kw["model_name"] != "dgcnn":   # (cfg3's dominant kernel is its kNN: knn_roofline below)
                    entry["roofline"] = {"bound": "hbm", "kernel": ROOFLINE_KERNEL, "achieved": entry["gemm_nt_GBps"],
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": entry["gemm_nt_GBps"] / HBM_PEAK_GBS,
                                         "traffic": None, "launches": launches,
                                         "note": "HIP events around the family's launches in the timed region of this run "
                                                 "(eager step, GEMMs of concurrent chains share the chip)"}
            if name.startswith("pn2_ssg_infer_published"):
                entry["vs_published"] = res["points_per_s"] / 35557.0
            if kw["model_name"] == "dgcnn" and not kw.get("graph"):
                entry["roofline"] = knn_roofline(kw["B"], kw["N"], device)
            out[name] = entry
            del run
        except BaseException as e:  # an extra must never cost the headline line
            if isinstance(e, KeyboardInterrupt):
                raise
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    from pointcloud_bridge_amd import rowmlp
    rowmlp.set_precision(args.precision)
    return out


# Dominant kernel of the step (profiles/): the fused row GEMM pcb_gemm_nt_* (forward and
# input-gradient GEMMs of every pointwise layer).  It is HBM-bound (skinny: K, N <= a few hundred).
# Work unit = one algorithmic HBM byte: every activation operand read once and the output written
# once; the library counts them per launch.
ROOFLINE_KERNEL = "pcb_gemm_nt_bf16"

if __name__ == "__main__":
    main()
