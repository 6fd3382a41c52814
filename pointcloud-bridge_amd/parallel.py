"""Data-parallel training over the GPUs of one node: one process per GPU, scenes sharded, one
gradient all-reduce per step over RCCL/xGMI.

The reference has no distributed code; every operator of the path is
per-scene, so the batch of scenes shards with no data-path collective.  The only exchange is the
gradient: 3-32 MB of fp32 per step, which is latency-bound on xGMI, so all gradients travel as ONE
flat bucket in ONE all-reduce (no per-parameter calls, no ring-size tuning).  Optionally the
BatchNorm statistics are synchronised too (torch.nn.SyncBatchNorm), which makes a G-GPU step
numerically the single-process step over the global batch.

Works with the `nccl` backend (= RCCL on ROCm) on GPUs and with `gloo` on CPU (used by the tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).

    Returns (rank, world_size, local_rank).  Single-process runs return (0, 1, 0) untouched."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not (os.environ.get("PCB_DIST_SINGLE") == "1" and "RANK" in os.environ):
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        # PCB_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box)
        backend = os.environ.get("PCB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def collectives(group=None):
    """True where the gradient exchange really calls the process group: more than one rank -- or a one-rank
    group under PCB_DIST_SINGLE=1, the rehearsal of the RCCL path on a one-GPU box (torch.distributed.run
    --nproc-per-node 1: communicator, streams, barriers and the captured step beside RCCL's watchdog thread are
    the real ones, only the peers are missing)."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("PCB_DIST_SINGLE") == "1"


def shard_scenes(num_scenes, rank, world):
    """Contiguous block of scene indices owned by `rank` (scenes are independent units)."""
    base, extra = divmod(num_scenes, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


class FlatGradAllReduce:
    """Averages the gradients of `params` across ranks with a single all-reduce.

    zero() drops the gradients (backward then writes fresh tensors: no accumulate kernels);
    reduce() packs them into ONE flat fp32 buffer with one concatenation (every parameter, zeros for
    a missing gradient: identical layout on every rank), all-reduces it once and hands each parameter
    a view of the averaged buffer.  With a single rank only the packing happens."""

    def __init__(self, params, group=None, keep_grad_tensors=False, assign_views=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        # keep_grad_tensors: the averaged values are copied back into the existing .grad tensors
        # instead of re-pointing .grad at the flat buffer (needed when a captured hipGraph writes
        # the gradients into fixed tensors on every replay)
        self.keep = keep_grad_tensors
        # assign_views=False: the caller consumes .flat itself (FlatAdam.step(bucket.flat)); the
        # per-parameter .grad tensors are left as they are (NOT averaged)
        self.assign_views = assign_views
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.coll = collectives(group)
        self.flat = None
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatGradAllReduce expects fp32 master parameters")

    def zero(self):
        for p in self.params:
            p.grad = None

    def pack(self):
        """One flat fp32 tensor of ALL parameters' gradients in parameter order -- zeros where a
        parameter has none this step, so every rank packs the same layout whatever branch it took."""
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])

    def reduce(self):
        """Sum over ranks, divide by the world size (mean, as DDP does).  With one rank: .flat is just
        the packed gradient (no collective), so callers need no special case."""
        self.flat = self.pack()
        if not self.coll:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        if not self.assign_views:
            return
        off = 0
        views = []
        for p in self.params:
            n = p.numel()
            views.append(self.flat[off:off + n].view_as(p))
            off += n
        if self.keep:
            live = [(p.grad, v) for p, v in zip(self.params, views) if p.grad is not None]
            torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
            for p, v in zip(self.params, views):
                if p.grad is None:
                    p.grad = v.clone()
        else:
            for p, v in zip(self.params, views):
                p.grad = v


# Streams besides the caller's on which backward nodes may run: a model that runs independent branches on
# streams of its own (models/pointnet2_utils.run_branches) registers them here.  A bucket is packed by the hook
# of whichever parameter's gradient lands last, on THAT node's stream; the other gradients of the bucket may
# have been produced on the other streams, so the packing waits for all of them first.
compute_streams = []


def register_compute_stream(stream):
    if all(stream != s for s in compute_streams):
        compute_streams.append(stream)


class OverlappedGradAllReduce:
    """The flat gradient all-reduce, started bucket by bucket WHILE the backward pass is still running.

    The parameters are split into buckets in parameter order (by default one per top-level child
    module: sa1, sa2, ..., the head).  Every parameter's gradient is copied into its slice of ONE
    flat fp32 buffer the moment autograd has produced it (post-accumulate-grad hook); when a
    bucket's last gradient has landed, its slice goes into an asynchronous all-reduce (RCCL runs it
    on its own stream: the encoder's gradients are still being computed while the decoder's travel).
    finish() waits for the collectives and returns the averaged flat gradient, laid out exactly like
    parallel.FlatAdam's parameter buffer.  A parameter that received no gradient in a step counts as
    zero (every rank then reduces the same layout).  With one rank the hooks only do the packing.

    ORDER.  Collectives must be issued in the same order on every rank.  A bucket is PACKED whenever its last gradient
    lands, but its all-reduce is only SENT once every bucket behind it in parameter order has been sent -- the backward
    pass walks the modules last to first, so that is the order buckets complete in anyway -- and finish() flushes what
    is left in the same order.  A rank on which a branch received no gradient (its bucket completes only in finish())
    therefore delays its own sends, it does not reorder them against its peers' (ADVICE r2)."""

    def __init__(self, params, buckets=None, group=None):
        self.params = [p for p in params if p.requires_grad]
        if any(p.dtype != torch.float32 for p in self.params):
            raise TypeError("OverlappedGradAllReduce expects fp32 master parameters")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.coll = collectives(group)
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        index = {id(p): i for i, p in enumerate(self.params)}
        if buckets is None:
            buckets = [self.params]
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.bucket_of = [None] * len(self.params)   # parameter -> bucket
        self.ranges = []                               # bucket -> (first float, one past the last, parameters)
        for b, plist in enumerate(buckets):
            ids = sorted(index[id(p)] for p in plist if id(p) in index)
            if not ids:
                continue
            if ids != list(range(ids[0], ids[-1] + 1)):
                raise ValueError("a bucket must be a contiguous run of the parameter list")
            k = len(self.ranges)
            for i in ids:
                self.bucket_of[i] = k
            lo = self.offsets[ids[0]]
            hi = self.offsets[ids[-1]] + self.params[ids[-1]].numel()
            self.ranges.append((lo, hi, len(ids)))
        if any(b is None for b in self.bucket_of):
            raise ValueError("every parameter must belong to a bucket")
        self.pending = [n for _, _, n in self.ranges]
        self.landed = [False] * len(self.params)
        self.packed = [False] * len(self.ranges)
        self.unsent = len(self.ranges)   # buckets [unsent, ...) have been sent; the next one to go is unsent - 1
        self.sent_order = []             # (tests) bucket numbers in the order their collectives were issued
        self.works = []
        self.pack_events = {}
        self.handles = [p.register_post_accumulate_grad_hook(self._hook(i)) for i, p in enumerate(self.params)]

    @classmethod
    def by_children(cls, module, group=None):
        """One bucket per top-level child of `module` (registration order = parameter order)."""
        params = [p for p in module.parameters() if p.requires_grad]
        seen, buckets = set(), []
        for child in module.children():
            plist = [p for p in child.parameters() if p.requires_grad and id(p) not in seen]
            seen.update(id(p) for p in plist)
            if plist:
                buckets.append(plist)
        rest = [p for p in params if id(p) not in seen]
        if rest:
            buckets.append(rest)
        return cls(params, buckets, group)

    def _hook(self, i):
        def hook(p):
            self.landed[i] = True
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        """Bucket b is complete: ONE concatenation packs its gradients into the flat buffer (a missing
        gradient counts as zero), then its slice starts travelling."""
        lo, hi, _ = self.ranges[b]
        members = [i for i, k in enumerate(self.bucket_of) if k == b]
        if self.flat.is_cuda and compute_streams:
            cur = torch.cuda.current_stream()
            for s in compute_streams:
                if s != cur:
                    cur.wait_stream(s)
            for i in members:
                if self.params[i].grad is not None:
                    self.params[i].grad.record_stream(cur)  # allocated on its own node's stream, read here
        torch.cat([(self.params[i].grad if self.params[i].grad is not None else torch.zeros_like(self.params[i])).reshape(-1)
                   for i in members], out=self.flat[lo:hi])
        for i in members:
            self.params[i].grad = None  # the flat buffer is the gradient from here on
        self.packed[b] = True
        while self.unsent > 0 and self.packed[self.unsent - 1]:
            self.unsent -= 1
            k = self.unsent
            self.sent_order.append(k)
            if self.coll:
                klo, khi, _ = self.ranges[k]
                if self.flat.is_cuda and k != b:
                    # packed earlier, by another node's hook, possibly on another stream: the collective (issued from
                    # the current stream) must see that packing
                    torch.cuda.current_stream().wait_event(self.pack_events[k])
                self.works.append(dist.all_reduce(self.flat[klo:khi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.coll and self.flat.is_cuda and not self.packed_sent(b):
            ev = self.pack_events.get(b)
            if ev is None:
                ev = self.pack_events[b] = torch.cuda.Event()
            ev.record()

    def packed_sent(self, b):
        return b >= self.unsent

    def zero(self):
        """Call before every backward pass."""
        for p in self.params:
            p.grad = None
        self.pending = [n for _, _, n in self.ranges]
        self.landed = [False] * len(self.params)
        self.packed = [False] * len(self.ranges)
        self.unsent = len(self.ranges)
        self.sent_order = []
        self.works = []

    def finish(self):
        """After backward: reduce what has not been reduced yet, wait, average.  Returns the flat gradient."""
        for i in reversed(range(len(self.landed))):   # (last bucket first: the order they are sent in)
            if not self.landed[i]:  # no gradient this step: counts as zeros, so that every rank reduces the same layout
                self.landed[i] = True
                b = self.bucket_of[i]
                self.pending[b] -= 1
                if self.pending[b] == 0:
                    self._launch(b)
        for w in self.works:
            w.wait()
        self.works = []
        if self.world > 1:
            self.flat.div_(self.world)
        return self.flat

    def close(self):
        for h in self.handles:
            h.remove()
        self.handles = []


class FlatAdam:
    """torch.optim.Adam (the reference's optimiser, train_MulSca_PN2.py:125: Adam with L2 weight
    decay) over ONE flat fp32 buffer: the parameters become views of it, the gradients arrive as one
    flat tensor (FlatGradAllReduce.flat, or one concatenation here), and the update is a single
    fused elementwise kernel instead of a multi-tensor launch chain over ~150 tensors.  The
    arithmetic per element is torch's fused Adam (torch._fused_adam_), so results match
    torch.optim.Adam(fused=True) bit for bit."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise TypeError("FlatAdam expects fp32 parameters on one device")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.flat = torch.cat([p.detach().reshape(-1) for p in self.params])
        off = 0
        for p in self.params:
            n = p.numel()
            p.data = self.flat[off:off + n].view_as(p)  # same values, storage inside the flat buffer
            off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_t = torch.zeros((), dtype=torch.float32, device=dev)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def step(self, flat_grad=None):
        """flat_grad: the gradients already packed in parameter order (FlatGradAllReduce.flat), else
        they are concatenated here.  Every parameter must have a gradient."""
        if flat_grad is None:
            if any(p.grad is None for p in self.params):
                raise RuntimeError("FlatAdam.step: a parameter has no gradient")
            flat_grad = torch.cat([p.grad.reshape(-1) for p in self.params])
        if flat_grad.numel() != self.flat.numel():
            raise ValueError("flat gradient does not match the parameters")
        self.step_t += 1
        # the parameters are views of self.flat: this update bumps no version counter of theirs, so the
        # eval-mode operand cache of the MLP engine is told explicitly
        from . import rowmlp
        rowmlp.note_parameter_update()
        torch._fused_adam_([self.flat], [flat_grad], [self.exp_avg], [self.exp_avg_sq], [], [self.step_t],
                           lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], weight_decay=self.weight_decay,
                           eps=self.eps, amsgrad=False, maximize=False)


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not collectives(group):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def sync_batchnorm(module, group=None):
    """Replace BatchNorm layers by SyncBatchNorm so statistics cover the global batch."""
    return torch.nn.SyncBatchNorm.convert_sync_batchnorm(module, group)
