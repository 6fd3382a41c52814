#!/usr/bin/env python
"""Headline benchmark: points/sec, forward + backward (+ gradient all-reduce + Adam step) of the
PointNet++ segmentation network on synthetic clouds, B=16 scenes x N=16384 points per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus 8 --steps 20 --warmup 5

One process per GPU; scenes are sharded (weak scaling: B scenes per GPU); the only collective is
one flat gradient all-reduce over RCCL.  Rank 0 prints ONE JSON line (contract in the task brief).
`roofline` is measured live with HIP events around the launches of the dominant HBM-bound kernel
inside the timed region; `cpu_baseline` times the ATen port of the reference's CPU path
(oracle/torch_port.py) on this host's cores over a bounded sample (rank 0, N=1 only).
"""
import argparse
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


def synthetic_batch(B, N, seed, device):
    """Unit-ball clouds normalised like utils/simpdataset.py:47-62, colours U[0,1), labels 0..4."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(B, N, 3, generator=g)
    p = v / v.norm(dim=-1, keepdim=True) * torch.rand(B, N, 1, generator=g) ** (1.0 / 3.0)
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)
    colors = torch.rand(B, N, 3, generator=g)
    labels = torch.randint(0, 5, (B, N), generator=g)
    return p.contiguous().to(device), colors.to(device), labels.to(device)


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
    raise ValueError(name)


def loss_fn(logits, labels, channel_dim):
    if channel_dim == 1:
        return F.cross_entropy(logits, labels)           # train_MulSca_PN2.py:161
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1))  # train_DGCNN.py:177-197


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
    n = max(1, min(5, int((budget_s - first) / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n
    return {"value": N / dt, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"{model_name} fwd+bwd, B=1 x N={N}, fp32, {n} timed steps after 1 warm-up "
                      f"({dt:.2f} s/step), oracle/torch_port.py (ATen port of the reference path)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="pn2_msg", choices=["pn2_msg", "pn2_ssg", "dgcnn", "bridgeseg"])
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU (default 16; dgcnn 8)")
    ap.add_argument("--npoints", type=int, default=None, help="points per scene (default 16384; dgcnn 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not overlap the next batch's FPS pyramid with the current backward pass")
    ap.add_argument("--sync-bn", action="store_true", help="SyncBatchNorm across ranks (fp32 mode)")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train: fwd+loss+bwd+all-reduce+Adam (the headline metric); infer: eval-mode forward only, "
                         "the reference's own published metric (eva_model.py:137-168, model_performance_comparison.csv)")
    ap.add_argument("--graph", action="store_true",
                    help="train mode: replay forward+backward as one captured hipGraph instead of launching every "
                         "kernel from the host (measured equal on MI355X: the step is GPU-bound, so off by default; "
                         "pn2_msg / pn2_ssg only)")
    ap.add_argument("--loss", default="ce", choices=["ce", "bridge"],
                    help="bridge: BridgeStructureLoss(alpha=80, rel_margin=0.3) of train_MulSca_BriStruNet_CB.py:151-156 "
                         "(bridgeseg / pn2_msg / pn2_ssg logits [B,C,N])")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="pointwise-MLP arithmetic: bf16 activations (BASELINE config 2) or the fp32 parity mode")
    args = ap.parse_args()

    from pointcloud_bridge_amd import ops, parallel, rowmlp
    rowmlp.set_precision(args.precision)
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP kernels, no CPU fallback)")
    local = local % torch.cuda.device_count()  # ranks may share a GPU in a gloo rehearsal
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    B = args.batch or (8 if args.model == "dgcnn" else 16)
    N = args.npoints or (8192 if args.model == "dgcnn" else 16384)

    torch.manual_seed(42)  # identical init on every rank; broadcast below makes it certain
    model, cdim = build_model(args.model)
    model = model.to(device).train()
    if args.sync_bn and world > 1:
        model = parallel.sync_batchnorm(model)
    parallel.broadcast_parameters(model)
    use_graph = args.mode == "train" and args.graph and args.precision == "bf16"
    if use_graph and args.model in ("dgcnn", "bridgeseg"):
        # replaying a captured DGCNN step ended in a GPU memory fault on MI355X (cause not found yet;
        # the eager step is clean under the same tests) -- refuse rather than risk the device; the
        # BridgeSeg step has not been captured yet
        raise SystemExit("--graph is supported for pn2_msg / pn2_ssg only")
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = parallel.FlatGradAllReduce(params, keep_grad_tensors=use_graph, assign_views=False)
    # train_MulSca_PN2.py:125: Adam(lr=1e-3, weight_decay=1e-4) -- as one fused update over a flat
    # parameter buffer (parallel.FlatAdam: torch's fused-Adam arithmetic, one launch instead of ~25)
    opt = parallel.FlatAdam(params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    xyz, colors, labels = synthetic_batch(B, N, 1000 + rank, device)
    torch.manual_seed(7 + rank)  # CPU generator: FPS start indices

    prefetch = (not args.no_prefetch) and hasattr(model, "prefetch")
    if args.loss == "bridge":
        if cdim != 1:
            raise SystemExit("--loss bridge expects [B,C,N] logits (bridgeseg, pn2_msg, pn2_ssg)")
        from pointcloud_bridge_amd.losses import BridgeStructureLoss
        crit = BridgeStructureLoss(alpha=80, rel_margin=0.3).to(device)  # train_MulSca_BriStruNet_CB.py:151-156

        def loss_of(logits):
            return crit(logits.float(), labels, xyz)
    else:
        def loss_of(logits):
            return loss_fn(logits, labels, cdim)

    def train_step():
        bucket.zero()
        loss = loss_of(model(xyz, colors))
        if prefetch:
            model.prefetch(xyz)  # sampling pyramid of the next batch, concurrent with this backward
        loss.backward()
        bucket.reduce()
        opt.step(bucket.flat)
        return loss

    def infer_step():
        with torch.no_grad():
            if prefetch and hasattr(model, "set_next"):
                model.set_next(xyz)  # pipelined serving: the next batch's FPS pyramid runs beside this pass's decoder
            return loss_of(model(xyz, colors))

    if args.mode == "infer":
        model.eval()
    step = train_step if args.mode == "train" else infer_step

    if use_graph:
        # Forward + loss + backward (+ the next batch's sampling pyramid on a side stream) replayed
        # as ONE hipGraph: ~700 kernel launches per step stop costing host time.  The CPU-generator
        # draws for FPS stay on the host (StaticSampling.draw), the gradient all-reduce (RCCL) and the
        # fused Adam step stay outside the graph.
        from pointcloud_bridge_amd.models import pointnet2_utils as pu
        static = None
        if hasattr(model, "prefetch"):
            static = pu.StaticSampling(xyz, [model.sa1.npoint, model.sa2.npoint, model.sa3.npoint])
            pu.set_static_sampling(static)
            static.draw()
            static.compute(xyz)
        side = torch.cuda.Stream()
        loss_buf = torch.zeros((), device=device)

        def fwd_bwd():
            loss = loss_of(model(xyz, colors))
            if static is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    static.compute(xyz)          # pyramid of the next batch, beside the backward pass
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
        with torch.cuda.graph(graph):
            fwd_bwd()

        grads = [p.grad for p in model.parameters() if p.grad is not None]  # the graph's fixed tensors

        def graph_step():
            if static is not None:
                static.draw()
            graph.replay()
            bucket.reduce()
            opt.step(bucket.flat)
            return loss_buf

        def eager_step():
            """The same step launched kernel by kernel (HIP events around the roofline kernel need
            real launches); gradients accumulate into the graph's tensors, zeroed first."""
            if static is not None:
                static.draw()
            torch._foreach_zero_(grads)
            fwd_bwd()
            bucket.reduce()
            opt.step(bucket.flat)
            return loss_buf

        step = graph_step

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # The interpreter's cyclic garbage collector stops the host for tens of milliseconds at a time
    # (a step creates thousands of short-lived tensor objects); such a pause drains the GPU queue.
    # As in production training loops it is run at chosen points instead: here, before and after
    # the timed steps.
    import gc
    gc.collect()
    gc.disable()
    fence()
    ops.kernel_timer_start()
    t0 = time.perf_counter()
    host_s = 0.0  # time the host spends enqueueing (diagnostic: host-bound vs GPU-bound)
    for i in range(args.steps):
        # HIP events around the roofline kernel on every 10th step (launched eagerly in graph mode)
        sampled = i % 10 == 0 and not os.environ.get("PCB_BENCH_NO_ROOFLINE")
        ops.kernel_timer_enable(sampled)
        h0 = time.perf_counter()
        loss = (eager_step if (use_graph and sampled) else step)()
        host_s += time.perf_counter() - h0
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    launches, kernel_ms, units = ops.kernel_timer_stop()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        # BASELINE.md publishes one number on this path that a bench mode reproduces exactly:
        # PointNet2 (SSG) inference, B=4 x N=4096, fp32, eval, 1 GPU (RTX 4090): 35 557 pts/s
        # (Highway_bridge/model_performance_comparison.csv:4).  The headline fwd+bwd metric has none.
        vs_baseline = None
        if (args.mode == "infer" and args.model == "pn2_ssg" and B == 4 and N == 4096
                and args.precision == "fp32" and world == 1):
            vs_baseline = (B * N / (dt / args.steps)) / 35557.0
        traffic = None  # HBM bytes per launch from the committed PMC passes of this same workload
        pmc = os.path.join(REPO, "profiles", "r01_pmc_gemm_nt_bf16.json")
        if (args.model == "pn2_msg" and args.precision == "bf16" and B == 16 and N == 16384
                and os.path.exists(pmc)):
            with open(pmc) as f:
                traffic = json.load(f)["traffic_bytes_per_launch"]
        alg_bytes = units * ROOFLINE_BYTES_PER_UNIT / max(launches, 1)  # per launch
        avg_s = kernel_ms / max(launches, 1) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if launches else 0.0
        out = {
            "metric": (("points/sec fwd+bwd, " if args.mode == "train" else "points/sec inference (eval forward), ")
                       + {"dgcnn": f"DGCNN k=20 EdgeConv seg N={N} B={B}",
                          "bridgeseg": f"BridgeSeg (EnhancedPointNet2: bridge encoders + PointNet++ MSG) N={N} B={B}"}
                       .get(args.model, f"PointNet++ seg N={N} B={B}")),
            "value": world * B * N / (dt / args.steps),
            "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": vs_baseline, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} {('fwd+' + ('CE' if args.loss == 'ce' else 'BridgeStructureLoss') + '+bwd+grad-allreduce+Adam') if args.mode == 'train' else ('eval-mode forward+CE' + (', next batch FPS pipelined' if (prefetch and hasattr(model, 'set_next')) else ''))}, B={B} scenes/GPU x N={N} pts, "
                                   f"unit-ball clouds (configs[1] of BASELINE.json)",
                       "scenes_per_gpu": B, "points_per_scene": N, "parallelism": f"dp{world} (scenes sharded)",
                       "loss": float(loss.detach()),
                       "host_enqueue_ms_per_step": host_s / args.steps * 1e3},
            "roofline": {"bound": "hbm", "kernel": ROOFLINE_KERNEL, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launches": launches, "avg_launch_us": avg_s * 1e6,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, N)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# Dominant kernel of the step (profiles/): the fused bf16 row GEMM pcb_gemm_nt_bf16 (forward and
# input-gradient GEMMs of every pointwise layer).  It is HBM-bound (skinny: K, N <= a few hundred).
# Work unit = one algorithmic HBM byte: every activation operand read once and the output written
# once, 2 B per bf16 element; the wrappers pass that count per launch.
ROOFLINE_KERNEL = "pcb_gemm_nt_bf16"
ROOFLINE_BYTES_PER_UNIT = 1.0

if __name__ == "__main__":
    main()
