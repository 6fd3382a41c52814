"""Drop-in for Highway_bridge/models/pointnet2_utils.py of UT-Team-Chun/Pointcloud-bridge.

Same public names, argument order, tensor layouts, int64 index dtype and state_dict keys as the
reference module, so `from models.pointnet2_utils import SetAbstraction, FeaturePropagation,
MultiScaleSetAbstraction, EnhancedFeaturePropagation` keeps working in the reference's containers
(models/model.py:10, models/pointnet2.py:8).  The neighbourhood operators run as hand-written
gfx950 kernels (..ops -> libpcb_hip.so); nothing of size [B,S,N] / [B,N,S] is materialised.

Inside a module the activations are kept channels-last ([rows, C]); the 1x1 convolutions of the
reference are evaluated as row GEMMs with the weights of the very same nn.Conv*/nn.BatchNorm*
sub-modules (mlp_convs.*, mlp_bns.*, conv_blocks.*, bn_blocks.*, attention.*, boundary_aware.*).
Tensors returned to the caller have the reference's shapes ([B,C,S] / [B,C,N]); they are
transposed views of channels-last storage.

GPU only: CPU tensors raise (there is no fallback path).
"""
import os
import threading
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, parallel, rowmlp


# ---------------------------------------------------------------------------------------------
# free functions (reference: pointnet2_utils.py:7-112)
# ---------------------------------------------------------------------------------------------
def square_distance(src, dst):
    """[B,N,3] x [B,M,3] -> [B,N,M] squared distances (reference :7-14), materialised."""
    return ops.square_distance(src, dst)


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,ns] (clamped to [0,N-1]) -> [B,S,(ns,)C] (reference :17-39)."""
    return ops.gather_rows(points, idx)


class SamplingState:
    """What a forward pass may pick up instead of computing it -- the prefetched FPS pyramid, parked ball-query / k-NN /
    inverted-index results and their double-set buffers, an installed StaticSampling pipeline, the scene shard of a
    strong-scaling run.  Every container model owns one (containers._SamplingPrefetchMixin.sampling) and makes it the
    current one for the duration of its calls (`sampling_scope`, thread-local), so two models in one process do not
    see each other's prefetches; the functions of this module called directly use the process default."""

    def __init__(self):
        self.prefetched = {}
        self.parked = {}   # coordinate-only results of prefetch_sampling other than the FPS pyramid: key -> (tensors, event, parity, source)
        self.owned = {}    # their storage: (kind, ordinal, parity) -> tensors, allocated once on the main stream
        self.parity = 0
        self.static = None
        self.scene_shard = None   # (rank, world) of this model's batches; None: the process default (set_scene_shard)


_default_state = SamplingState()
_tls = threading.local()
_process_shard = (0, 1)


def _st():
    return getattr(_tls, "state", None) or _default_state


class sampling_scope:
    """with sampling_scope(state): the module's lookups and prefetches use `state` (nests; per thread)."""

    def __init__(self, state):
        self.state = state

    def __enter__(self):
        self.prev = getattr(_tls, "state", None)
        _tls.state = self.state
        return self.state

    def __exit__(self, *exc):
        _tls.state = self.prev
        return False


def set_scene_shard(rank, world):
    """Data-parallel runs that split ONE global batch of scenes over `world` ranks (strong scaling):
    every rank draws the start indices of the whole global batch from its CPU generator -- seeded
    identically on all ranks -- and keeps the entries of its own scenes.  The sharded run then
    consumes the reference's RNG stream (pointnet2_utils.py:69) exactly like a single process over the
    global batch, so its samples, and with SyncBatchNorm its step, equal the single-process step.
    This sets the PROCESS default (one process = one rank); a model can carry its own in
    `model.sampling.scene_shard`, and farthest_point_sample takes `shard=` explicitly."""
    global _process_shard
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    _process_shard = (int(rank), int(world))


def scene_shard(state=None):
    st = state if state is not None else _st()
    return st.scene_shard if st.scene_shard is not None else _process_shard


def farthest_point_sample(xyz, npoint, shard=None):
    """xyz [B,N,3] -> [B,npoint] int64 (reference :63-80).

    Consumes exactly one torch.randint(0, N, (B,)) from the CPU default generator, like the
    reference (:69), so seeded runs sample the same start points (see set_scene_shard for sharded
    global batches)."""
    B, N, _ = xyz.shape
    ops._need_cuda(xyz)  # GPU only: fail here, before any staging copy
    rank, world = shard if shard is not None else scene_shard()
    start = torch.randint(0, N, (B * world,), dtype=torch.long)[rank * B:(rank + 1) * B]
    # pinned staging + asynchronous copy: a pageable .to(device) would stall the host until the
    # stream has drained, three times per forward pass
    start = start.pin_memory().to(xyz.device, non_blocking=True)
    return ops.furthest_point_sample(xyz, npoint, start)


# ---------------------------------------------------------------------------------------------
# sampling prefetch: FPS depends on the coordinates only (not on any weight), so the sampling pyramid
# of the NEXT batch can run on a side stream (it occupies one CU per scene) while the current batch
# is still in its backward pass -- the GPU-side counterpart of the reference's DataLoader workers
# preparing the next batch (train_MulSca_PN2.py:92-106).  Results and RNG consumption are unchanged:
# the start indices are drawn from the CPU generator in the same order, just earlier.
# ---------------------------------------------------------------------------------------------
_side_stream = None


def _key(xyz, npoint):
    return (xyz.data_ptr(), xyz._version, tuple(xyz.shape), int(npoint))


# Parked results are looked up by the ADDRESS of the coordinate tensor they were computed from.  An
# address alone can be inherited: if batch A is dropped without being forwarded (last step of an
# epoch, an exception, eval starting), the caching allocator readily hands A's block to the next
# tensor of the same shape -- which would then pick up A's samples and neighbours.  So every entry
# also holds a weak reference to its level-0 tensor and is only served while that very tensor object
# is the one asking (or, for the coarser levels, while it is still alive: their coordinate tensors
# are owned by the entries themselves, so their addresses cannot be reused meanwhile).
class _Source:
    """Identity of the level-0 coordinate tensor a prefetch was issued for."""

    def __init__(self, xyz):
        self.ref = weakref.ref(xyz)
        self.ptr = xyz.data_ptr()

    def alive(self):
        return self.ref() is not None

    def matches(self, xyz):
        """True unless `xyz` sits at the source's address without being the source."""
        return xyz.data_ptr() != self.ptr or self.ref() is xyz


def _drop_dead():
    """Forget everything parked for coordinate tensors that no longer exist."""
    st = _st()
    for table in (st.prefetched, st.parked):
        for k in [k for k, v in table.items() if not v[-1].alive()]:
            del table[k]


def side_stream(device):
    """The stream the coordinate-only work of the NEXT batch runs on (FPS pyramid here, the
    neighbourhood geometry of attention_modules.BridgeStructureEncoding)."""
    global _side_stream
    if _side_stream is None:
        _side_stream = torch.cuda.Stream(device=device)
    return _side_stream


def prefetch_sampling(xyz, npoints, balls=None, propagation=None):
    """Compute the FPS pyramid xyz -> npoints[0] -> npoints[1] -> ... on a side stream and park it
    for the next forward pass over these coordinates (SetAbstraction / MultiScaleSetAbstraction pick
    it up by tensor identity).  Call after the forward of the current batch.

    The other coordinate-only operators of a PointNet++ pass can ride along:
      balls[l]    = (radii, nsamples) of the set abstraction at level l  -> its ball-query indices
      propagation = [(fine level, coarse level, k), ...] with level 0 = xyz, level l = the l-th
                    sampled cloud                                         -> the k-NN of each decoder stage
    Their results (megabytes of indices) are copied into buffers this module owns -- two sets, used
    alternately, because the backward pass of the current batch still reads the set the current
    forward pass took -- so that no large block changes streams in the caching allocator.  They are
    parked under the identity of the level tensors, which the forward pass receives again from
    _sample, and picked up by _ball_indices / _nearest.  Contract: a parked result (and whatever
    autograd saved of it) is valid until the SECOND prefetch after the one that produced it, i.e. for
    the usual forward -> prefetch -> backward -> step loop and for pipelined inference."""
    st = _st()
    side_stream(xyz.device)
    main = torch.cuda.current_stream()
    _side_stream.wait_stream(main)
    src = _Source(xyz)
    st.prefetched.clear()
    # entries of the batch in flight stay (a pipelined inference pass starts this prefetch before its
    # own decoder has taken its k-NN); their set of buffers is not the one written now
    for stale in [k for k, v in st.parked.items() if v[2] != st.parity]:
        del st.parked[stale]
    st.parity ^= 1
    cur = xyz
    levels = [xyz]

    def park(key, res):
        slot = (key[0], sum(1 for v in st.parked.values() if v[2] == st.parity), st.parity)
        held = st.owned.get(slot)
        if held is None or any(h.shape != r.shape or h.dtype != r.dtype for h, r in zip(held, res)):
            with torch.cuda.stream(main):  # blocks of the main stream's pool, for good
                held = st.owned[slot] = tuple(torch.empty_like(r) for r in res)
        for h, r in zip(held, res):
            h.copy_(r)
        ev = torch.cuda.Event()
        ev.record(_side_stream)
        st.parked[key] = (held, ev, st.parity, src)

    with torch.cuda.stream(_side_stream):
        for l, npoint in enumerate(npoints):
            idx = farthest_point_sample(cur, npoint)
            new_xyz = index_points(cur, idx)
            ev = torch.cuda.Event()
            ev.record(_side_stream)
            st.prefetched[_key(cur, npoint)] = (idx, new_xyz, ev, src)
            if balls is not None and balls[l] is not None:
                radii, nsamples = balls[l]
                park(_ball_key(radii, nsamples, cur, new_xyz), _ball_indices_now(radii, nsamples, cur, new_xyz))
            cur = new_xyz
            levels.append(new_xyz)
        for fine, coarse, k in (propagation or ()):
            nn_key = ("nn", levels[fine].data_ptr(), levels[coarse].data_ptr(), int(k))
            park(nn_key, ops.three_nn(levels[fine], levels[coarse], k))
            if torch.is_grad_enabled():
                # the inverted index the interpolation's backward pass reduces over
                held_idx = st.parked[nn_key][0][1]
                park(("csr", held_idx.data_ptr()), rowmlp.build_interp_csr(held_idx, levels[coarse].shape[1]))
        ev = torch.cuda.Event()
        ev.record(_side_stream)
    # one FPS workgroup per scene; the GEMMs beside it leave twice that many CUs alone (measured)
    ops.set_background_work(ev, 2 * xyz.shape[0])


class StaticSampling:
    """The coordinate-only results of a step in persistent device buffers, for steps replayed from a hipGraph.

    A captured step cannot draw from the CPU generator or allocate.  So the FPS start indices live in static
    device tensors that `draw()` refreshes on the host before every replay (same torch.randint calls, same order
    as farthest_point_sample), and `compute(xyz_next)` enqueues -- inside the captured step, on a side stream
    next to the backward pass -- the FPS pyramid of the NEXT batch and, when given, what else depends on
    coordinates only (`balls`, `propagation` as in prefetch_sampling: ball-query indices per level, the k-NN and
    the inverted index of each decoder stage).

    Two sets of buffers: compute() writes the STAGING set, commit() -- the first thing a captured step does --
    copies it into the LIVE set the modules read.  The backward pass of a step still reads the live set (centroid
    coordinates for the gradient of the coordinate weights, neighbour indices for the scatter) while the side
    stream already produces the next step's results: with one set those reads raced with that work.
    SetAbstraction / MultiScaleSetAbstraction / the propagation stages read the live buffers (see _sample,
    _ball_indices, _nearest, _interp_csr) while a pipeline is installed with set_static_sampling()."""

    def __init__(self, xyz, npoints, balls=None, propagation=None, jobs=None):
        B, N, _ = xyz.shape
        dev = xyz.device
        self.levels = []
        self.balls = balls
        self.propagation = list(propagation or ())
        # further coordinate-only work of the model: (key, level, fn) -- fn(cloud of that level) -> tuple of tensors,
        # computed with the pyramid, staged and committed like the rest, found by lookup_job(key, cloud)
        self.jobs = list(jobs or ())
        self.extra = {}   # key -> (live tensors, staging tensors), allocated at the first compute()
        self.state = _st()  # the SamplingState this pipeline serves (a model's own when built by model.static_sampling)
        self._pending = []  # (staging, result) pairs of the compute() under way
        n_in = N
        # FPS start indices of all levels: ONE device buffer, filled by one asynchronous copy per draw() from a ring of
        # pinned host buffers.  The host runs ahead of the GPU (a replay is enqueued in ~4 ms, executes in ~7): a single
        # pinned buffer would be rewritten by the next draw() before the previous copy has executed, and a step would
        # sample from its successor's start indices -- every slot carries an event recorded behind its copy
        # and is only rewritten once that event has completed.
        self._start_all = torch.zeros(len(npoints), B, dtype=torch.long, device=dev)
        self._pinned = [torch.zeros(len(npoints), B, dtype=torch.long).pin_memory() for _ in range(4)]
        self._pin_events = [None] * len(self._pinned)
        self._pin_next = 0
        for l, s in enumerate(npoints):
            lv = {"n_in": n_in, "npoint": int(s), "start": self._start_all[l]}
            for tag in ("", "_s"):
                lv["idx" + tag] = torch.zeros(B, int(s), dtype=torch.long, device=dev)
                lv["new_xyz" + tag] = torch.zeros(B, int(s), 3, dtype=torch.float32, device=dev)
            self.levels.append(lv)
            n_in = int(s)

    def draw(self):
        """Host side, outside any capture: one torch.randint per level from the CPU generator
        (reference :69), staged through pinned memory into the static start buffers."""
        i = self._pin_next
        self._pin_next = (i + 1) % len(self._pinned)
        if self._pin_events[i] is None:
            self._pin_events[i] = torch.cuda.Event()
        else:
            self._pin_events[i].synchronize()   # the copy that last read this slot has executed (normally long ago)
        host = self._pinned[i]
        rank, world = scene_shard(self.state)
        for l, lv in enumerate(self.levels):
            B = host.shape[1]
            host[l].copy_(torch.randint(0, lv["n_in"], (B * world,), dtype=torch.long)[rank * B:(rank + 1) * B])
        self._start_all.copy_(host, non_blocking=True)
        self._pin_events[i].record()

    def _stage(self, key, res):
        held = self.extra.get(key)
        if held is None:  # first call: outside the capture (the warm-up steps)
            held = self.extra[key] = (tuple(torch.empty_like(r) for r in res), tuple(torch.empty_like(r) for r in res))
        # (the copies of one compute() go out together: _flush_stage, one launch instead of two dozen copy nodes)
        self._pending.extend((h, r) for h, r in zip(held[1], res))

    def _flush_stage(self):
        pend, self._pending = self._pending, []
        if not pend:
            return
        ok = all(h.is_contiguous() and r.is_contiguous() and h.dtype == r.dtype and h.shape == r.shape
                 and (h.numel() * h.element_size()) % 4 == 0 and h.numel() > 0 for h, r in pend)
        if not ok:
            for h, r in pend:
                h.copy_(r)
            return
        import ctypes
        n = len(pend)
        arr = ctypes.c_longlong * n
        dst = arr(*[h.data_ptr() for h, _ in pend])
        src = arr(*[r.data_ptr() for _, r in pend])
        nbytes = arr(*[h.numel() * h.element_size() for h, _ in pend])
        with ops.on_device(pend[0][0].device):
            ops._launch("pcb_copy_list", n, ctypes.addressof(dst), ctypes.addressof(src), ctypes.addressof(nbytes), n)

    def compute(self, xyz):
        """Enqueue everything for `xyz` (the NEXT step's coordinates) on the current stream, into the staging set."""
        cur = xyz
        clouds = [xyz]
        for l, lv in enumerate(self.levels):
            ops.furthest_point_sample_into(cur, lv["start"], lv["idx_s"])
            ops.gather_rows_into(cur, lv["idx_s"], lv["new_xyz_s"])
            if self.balls is not None and self.balls[l] is not None:
                radii, nsamples = self.balls[l]
                self._stage(("ball", l), _ball_indices_now(radii, nsamples, cur, lv["new_xyz_s"]))
            cur = lv["new_xyz_s"]
            clouds.append(cur)
        for key, level, fn in self.jobs:
            self._stage(("job", key), tuple(fn(clouds[level])))
        for fine, coarse, k in self.propagation:
            nn_res = ops.three_nn(clouds[fine], clouds[coarse], k)
            self._stage(("nn", fine, coarse, int(k)), nn_res)
            if torch.is_grad_enabled():
                self._stage(("csr", fine, coarse, int(k)), rowmlp.build_interp_csr(nn_res[1], clouds[coarse].shape[1]))
        self._flush_stage()

    def compute_beside(self, xyz, calls=1 << 30):
        """compute(xyz) on this pipeline's own side stream, forked from the current stream (pipelined inference
        inside a captured pass: the next batch's sampling beside the decoder); join() brings it back.  The forward
        stacks that follow leave the pyramid's CUs alone (ops.set_background_work)."""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            self.compute(xyz)
        self._forked = True
        ops.set_background_work(torch.cuda.Event(), 2 * xyz.shape[0], calls=calls)

    def join(self):
        if getattr(self, "_forked", False):
            torch.cuda.current_stream().wait_stream(self._side)
            self._forked = False

    def commit(self):
        """staging -> live (capturable; the first operation of a captured step): ONE launch that walks a table of
        all (live, staging) pairs -- two dozen copy nodes at the top of every replay otherwise.  The table is built
        at the first call after the buffers exist (the eager warm-up steps, outside any capture)."""
        pairs = []
        for lv in self.levels:
            pairs += [(lv["idx"], lv["idx_s"]), (lv["new_xyz"], lv["new_xyz_s"])]
        for live, staged in self.extra.values():
            pairs += list(zip(live, staged))
        sig = tuple((d.data_ptr(), r.data_ptr(), d.numel() * d.element_size()) for d, r in pairs)
        if getattr(self, "_commit_sig", None) != sig:
            if torch.cuda.is_current_stream_capturing():
                for d, r in pairs:        # (a buffer appeared during the capture itself: plain copies)
                    d.copy_(r)
                return
            vals, blocks = [], 0
            for dptr, rptr, nbytes in sig:
                vals += [dptr, rptr, nbytes, blocks]
                blocks += (nbytes + 16383) // 16384
            self._commit_table = torch.tensor(vals, dtype=torch.int64, device=pairs[0][0].device)
            self._commit_sig, self._commit_blocks = sig, blocks
        with ops.on_device(pairs[0][0].device):
            ops._launch("pcb_copy_table", len(pairs), self._commit_table.data_ptr(), len(pairs), self._commit_blocks)

    def _level_of(self, t):
        """0 for a level-0 cloud (any tensor of the input size), l for the live coordinates of level l, else None."""
        for l, lv in enumerate(self.levels):
            if t.data_ptr() == lv["new_xyz"].data_ptr():
                return l + 1
        return 0 if t.shape[1] == self.levels[0]["n_in"] else None

    def lookup(self, xyz, npoint):
        prev = None
        for lv in self.levels:
            if lv["npoint"] == int(npoint) and (prev is None or xyz.data_ptr() == prev["new_xyz"].data_ptr()):
                if xyz.shape[1] == lv["n_in"]:
                    return lv["idx"], lv["new_xyz"]
            prev = lv
        return None

    def lookup_ball(self, radii, nsamples, xyz, new_xyz):
        l = self._level_of(new_xyz)
        if l is None or l == 0 or self.balls is None or self.balls[l - 1] is None:
            return None
        r0, n0 = self.balls[l - 1]
        if tuple(float(r) for r in r0) != tuple(float(r) for r in radii) or tuple(int(n) for n in n0) != tuple(int(n) for n in nsamples):
            return None
        held = self.extra.get(("ball", l - 1))
        return None if held is None else held[0]

    def lookup_job(self, key, xyz):
        """The live results of job `key` if `xyz` is the live cloud of the job's level, else None."""
        for k, level, _ in self.jobs:
            if k == key and self._level_of(xyz) == level:
                held = self.extra.get(("job", key))
                return None if held is None else held[0]
        return None

    def lookup_extra(self, kind, xyz1, xyz2, k):
        f, c = self._level_of(xyz1), self._level_of(xyz2)
        held = self.extra.get((kind, f, c, int(k))) if f is not None and c is not None else None
        return None if held is None else held[0]


_static_states = []   # the states a pipeline is installed in (set_static_sampling(None) removes them all)


def set_static_sampling(pipeline):
    """Install a StaticSampling pipeline for subsequent forward passes -- in the SamplingState it was built under
    (model.static_sampling(...) builds it under the model's own) -- or, with None, remove every installed one."""
    if pipeline is None:
        for st in _static_states:
            st.static = None
        _static_states.clear()
        _st().static = None
        return
    st = pipeline.state
    st.static = pipeline
    if st not in _static_states:
        _static_states.append(st)


def static_sampling():
    return _st().static


def _sample(xyz, npoint):
    """(fps_idx, new_xyz) for this level: static buffers of a captured step, else the prefetched
    pair if there is one, else computed now."""
    st = _st()
    if st.static is not None:
        hit = st.static.lookup(xyz, npoint)
        if hit is not None:
            return hit
    hit = st.prefetched.pop(_key(xyz, npoint), None) if st.prefetched else None
    if hit is not None and not (hit[3].alive() and hit[3].matches(xyz)):
        _drop_dead()
        hit = None
    if hit is None:
        idx = farthest_point_sample(xyz, npoint)
        return idx, index_points(xyz, idx)
    idx, new_xyz, ev, _ = hit
    main = torch.cuda.current_stream()
    main.wait_event(ev)
    idx.record_stream(main)
    new_xyz.record_stream(main)
    return idx, new_xyz


def _take_parked(key, xyz=None):
    """The parked result for `key`, if its prefetch was issued for a coordinate tensor that is still
    alive (and, when the key is built from `xyz`'s address, for that very tensor)."""
    hit = _st().parked.pop(key, None)
    if hit is None:
        return None
    res, ev, _, src = hit
    if not src.alive() or (xyz is not None and not src.matches(xyz)):
        _drop_dead()
        return None
    torch.cuda.current_stream().wait_event(ev)
    return res


def _ball_key(radii, nsamples, xyz, new_xyz):
    return ("ball", xyz.data_ptr(), new_xyz.data_ptr(), tuple(float(r) for r in radii), tuple(int(n) for n in nsamples))


def _ball_indices_now(radii, nsamples, xyz, new_xyz):
    if len(radii) == 2:
        return tuple(ops.ball_query2(list(radii), list(nsamples), xyz, new_xyz))  # two radii, one sweep
    return tuple(ops.ball_query(r, ns, xyz, new_xyz) for r, ns in zip(radii, nsamples))


def _ball_indices(radii, nsamples, xyz, new_xyz):
    """Ball-query indices of every scale of a set abstraction: parked by prefetch_sampling or computed now."""
    st = _st()
    if st.static is not None:
        hit = st.static.lookup_ball(radii, nsamples, xyz, new_xyz)
        if hit is not None:
            return hit
    hit = _take_parked(_ball_key(radii, nsamples, xyz, new_xyz), xyz) if st.parked else None
    return hit if hit is not None else _ball_indices_now(radii, nsamples, xyz, new_xyz)


def _nearest(xyz1, xyz2, k):
    """(d2, idx) of the k nearest xyz2 points of every xyz1 point: parked by prefetch_sampling or computed now."""
    st = _st()
    if st.static is not None:
        hit = st.static.lookup_extra("nn", xyz1, xyz2, k)
        if hit is not None:
            return hit
    hit = _take_parked(("nn", xyz1.data_ptr(), xyz2.data_ptr(), int(k)), xyz1) if st.parked else None
    return hit if hit is not None else ops.three_nn(xyz1, xyz2, k)


def query_ball_point(radius, nsample, xyz, new_xyz):
    """-> [B,S,nsample] int64, first nsample in-radius indices in ascending order (reference :97-112)."""
    return ops.ball_query(radius, nsample, xyz, new_xyz)


def sample_and_group(npoint, radius, nsample, xyz, points):
    """FPS + ball query + grouping (reference :42-60).

    xyz [B,N,3], points [B,N,C] or None -> new_xyz [B,S,3], new_points [B,S,nsample,3+C]
    with the centred coordinates FIRST (:56)."""
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    return new_xyz, ops.group_points(xyz, new_xyz, points, idx)


# ---------------------------------------------------------------------------------------------
# pointwise layers on channels-last rows: ..rowmlp (fp32 parity mode or bf16 fused-kernel mode)
# ---------------------------------------------------------------------------------------------
def _channels_last(points):
    """[B,C,N] (any strides) -> contiguous [B,N,C]; free when `points` came out of this module."""
    return points.transpose(1, 2).contiguous()


def _seq_rows(seq, x):
    """Run an nn.Sequential of Conv1d / BatchNorm1d / ReLU / Sigmoid on rows."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (nn.Conv1d, nn.Conv2d)):
            if i + 1 < len(mods) and isinstance(mods[i + 1], (nn.BatchNorm1d, nn.BatchNorm2d, nn.SyncBatchNorm)):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                x = rowmlp.conv_bn_act(m, mods[i + 1], x, rowmlp.ACT_RELU if relu else rowmlp.ACT_NONE)
                i += 3 if relu else 2
                continue
            x = rowmlp.conv_rows(m, x)
        elif isinstance(m, nn.ReLU):
            x = F.relu(x)
        elif isinstance(m, nn.Sigmoid):
            x = torch.sigmoid(x)
        elif isinstance(m, nn.Dropout):
            x = m(x)
        else:
            raise TypeError(f"unsupported layer in pointwise stack: {type(m).__name__}")
        i += 1
    return x


# ---------------------------------------------------------------------------------------------
# modules (reference: pointnet2_utils.py:115-360)
# ---------------------------------------------------------------------------------------------
def _grouped_mlp(convs, bns, xyz, new_xyz, feat, idx):
    """Grouping + shared MLP + max over the neighbours of one set-abstraction scale
    (reference :51-58 + :149-154 / :342-356) -> [B*S, C] rows.

    bf16 rows with differentiable features of useful width: the first 1x1 conv is linear in
    [x_j - c_s | f_j], so its feature part is evaluated per point (u = F Wf^T on N rows,
    rowmlp.point_linear) and the rows y0[s,j] = u[idx[s,j]] + Wx (x_j - c_s) are gathered
    (rowmlp.gathered_mlp; the coordinate difference is formed in fp32 as in the reference): the
    grouped tensor is never written and the first GEMM, its input gradient and the scatter of a
    (3+C)-wide row gradient shrink to C0-wide gathers/scatters.  Otherwise rows are grouped first."""
    B, S, ns = idx.shape
    cf = 0 if feat is None else feat.shape[2]
    # (with gradients off -- inference -- the gathered form is simply the cheaper forward)
    if (feat is not None and (feat.requires_grad or not torch.is_grad_enabled()) and cf >= 32 and cf % 8 == 0
            and rowmlp.gathered_ok(convs, bns)):
        N = xyz.shape[1]
        w0 = convs[0].weight.view(convs[0].out_channels, 3 + cf)
        wx, wf = rowmlp.split_cols(w0, 3)
        u = rowmlp.point_linear(feat.reshape(B * N, cf), wf)
        return rowmlp.gathered_mlp(convs, bns, u, None, idx, pool=ns, wx=wx, xyz=xyz, ctr=new_xyz)
    rows, perm = rowmlp.group_rows(xyz, new_xyz, feat, idx)
    return rowmlp.mlp_rows(convs, bns, rows, pool=ns, perm=perm)



class SetAbstraction(nn.Module):
    """Single-scale set abstraction (reference :115-156)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        widths = [in_channel] + list(mlp)
        for cin, cout in zip(widths[:-1], widths[1:]):
            self.mlp_convs.append(nn.Conv2d(cin, cout, 1))
            self.mlp_bns.append(nn.BatchNorm2d(cout))

    def forward(self, xyz, points):
        """xyz [B,N,3], points [B,C,N] or None -> new_xyz [B,S,3], new_points [B,mlp[-1],S]."""
        feat = None if points is None else _channels_last(points)
        B = xyz.shape[0]
        _, new_xyz = _sample(xyz, self.npoint)
        idx, = _ball_indices([self.radius], [self.nsample], xyz, new_xyz)
        x = _grouped_mlp(self.mlp_convs, self.mlp_bns, xyz, new_xyz, feat, idx)
        return new_xyz, x.view(B, self.npoint, -1).transpose(1, 2)


# Independent sub-chains of a module (the scales of an MSG set abstraction, the boundary term of a decoder
# stage, the levels of the feature fusion) are chains of small DEPENDENT launches, and a small launch
# between two large ones costs ~6 us of an otherwise idle chip (tools/micro/launch_floor.hip).  Run on
# streams of their own, one chain's gaps are filled by another chain's kernels; autograd replays every
# backward node on the stream of its forward, so the backward pass overlaps the same way.
# PCB_BRANCH_STREAMS=0 turns it off (every chain on the caller's stream).
_branch_streams_on = os.environ.get("PCB_BRANCH_STREAMS", "1") == "1"
_branch_pool = []


def set_branch_streams(flag):
    global _branch_streams_on
    _branch_streams_on = bool(flag)


def branch_streams_enabled():
    return _branch_streams_on


def _branches(device, n):
    while len(_branch_pool) < n:
        _branch_pool.append(torch.cuda.Stream(device=device))
        parallel.register_compute_stream(_branch_pool[-1])  # gradient buckets packed during backward wait for it
    return _branch_pool[:n]


def run_branches(device, fns, inputs=()):
    """[fn() for fn in fns], fn i > 0 on branch stream i - 1 (forked from and joined to the current stream).
    Every fn returns a tensor; the results are safe to use on the current stream afterwards.
    inputs: the tensors of the CURRENT stream the branches read (and may save for backward).  In the backward
    pass a branch reads them on its own stream and drops them with no join behind it, so they are marked as
    used by the branch streams -- their blocks are not handed out again before that work is done."""
    # (a no-grad pass is host-bound -- 3 ms of enqueueing for 2.7 ms of kernels at B=16 x N=16384 -- and the
    # fork/join calls only add to that: branches are for training steps)
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if (not (_branch_streams_on and device.type == "cuda" and len(fns) > 1 and torch.is_grad_enabled())
            or (capturing and os.environ.get("PCB_BRANCH_IN_CAPTURE", "0") != "1")):
        return [fn() for fn in fns]
    main = torch.cuda.current_stream()
    streams = [main] + _branches(device, len(fns) - 1)
    for st in streams[1:]:
        st.wait_stream(main)
        for t in inputs:
            if t is not None and t.is_cuda and not capturing:
                t.record_stream(st)
    outs = []
    for st, fn in zip(streams, fns):
        with torch.cuda.stream(st):
            outs.append(fn())
    for st, o in zip(streams[1:], outs[1:]):
        main.wait_stream(st)
        if not capturing:
            o.record_stream(main)
    return outs


class MultiScaleSetAbstraction(nn.Module):
    """Multi-scale grouping set abstraction (reference :302-360).  `in_channel` already counts the
    3 centred coordinates (models/model.py:69,75-76)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp):
        super().__init__()
        self.npoint = npoint
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        widths = [in_channel] + list(mlp)
        for _ in radius_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            for cin, cout in zip(widths[:-1], widths[1:]):
                convs.append(nn.Conv2d(cin, cout, 1))
                bns.append(nn.BatchNorm2d(cout))
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points):
        """xyz [B,N,3], points [B,C,N] or None -> new_xyz [B,S,3], [B, len(radius)*mlp[-1], S]."""
        feat = None if points is None else _channels_last(points)
        _, new_xyz = _sample(xyz, self.npoint)  # one FPS for all scales (:335)
        idx_list = _ball_indices(self.radius_list, self.nsample_list, xyz, new_xyz)
        B = xyz.shape[0]
        # the scales are independent chains: each on its own stream (run_branches)
        outs = run_branches(xyz.device, [
            (lambda i=i, idx=idx: _grouped_mlp(self.conv_blocks[i], self.bn_blocks[i], xyz, new_xyz, feat, idx)
             .view(B, self.npoint, -1)) for i, idx in enumerate(idx_list)], inputs=(xyz, new_xyz, feat, *idx_list))
        return new_xyz, torch.cat(outs, dim=2).transpose(1, 2)


def _interpolate(xyz1, xyz2, points2, k):
    """points2 [B,D,S] at xyz2 -> [B,N,D] at xyz1 by inverse-distance weights of the k nearest."""
    S = xyz2.shape[1]
    if S == 1:
        # The reference's S == 1 branch (:181-182 / :250-251) builds a [B,D,N] tensor where the
        # following cat / conv need [B,N,D] and raises RuntimeError for every input; there is no
        # reference result to reproduce, so this raises as well.
        raise RuntimeError("feature propagation from a single centroid (S == 1) fails in the reference "
                           "(shape mismatch in its repeat branch) and is not supported")
    d2, idx = _nearest(xyz1, xyz2, k)
    feat = _channels_last(points2)
    out = ops.three_interpolate(feat.float(), d2, idx)
    return out.to(feat.dtype)


def _propagate_rows(xyz1, xyz2, points1, points2, k):
    """Rows [B*N, *] = cat([points1, interpolate(points2)]) (skip features FIRST, reference :201 / :272)
    plus the column-layout code `perm` of rowmlp (0 = reference order)."""
    B, N, _ = xyz1.shape
    S, C = xyz2.shape[1], points2.shape[1]
    if S > 1 and C % rowmlp.mode().q == 0:
        d2, idx = _nearest(xyz1, xyz2, k)
        st = _st()
        csr = st.static.lookup_extra("csr", xyz1, xyz2, k) if st.static is not None else None
        if csr is None:
            csr = _take_parked(("csr", idx.data_ptr())) if st.parked else None
        skip = None if points1 is None else _channels_last(points1).reshape(B * N, -1)
        return rowmlp.interpolate_concat(skip, _channels_last(points2), d2, idx, csr)
    x = _interpolate(xyz1, xyz2, points2, k)
    if points1 is not None:
        x = torch.cat([_channels_last(points1).to(x.dtype), x], dim=-1)
    return x.reshape(B * N, -1), 0


class FeaturePropagation(nn.Module):
    """3-NN inverse-distance feature propagation (reference :159-211)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        widths = [in_channel] + list(mlp)
        for cin, cout in zip(widths[:-1], widths[1:]):
            self.mlp_convs.append(nn.Conv1d(cin, cout, 1))
            self.mlp_bns.append(nn.BatchNorm1d(cout))

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,N,3], xyz2 [B,S,3], points1 [B,D1,N] or None, points2 [B,D2,S] -> [B,mlp[-1],N]."""
        B, N, _ = xyz1.shape
        x, perm = _propagate_rows(xyz1, xyz2, points1, points2, 3)
        x = rowmlp.mlp_rows(self.mlp_convs, self.mlp_bns, x, perm=perm)
        return x.view(B, N, -1).transpose(1, 2)


class EnhancedFeaturePropagation(nn.Module):
    """4-NN propagation + channel attention + boundary term + residual (reference :214-298)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.attention = nn.Sequential(
            nn.Conv1d(in_channel, in_channel // 4, 1),
            nn.BatchNorm1d(in_channel // 4),
            nn.ReLU(),
            nn.Conv1d(in_channel // 4, in_channel, 1),
            nn.Sigmoid())
        self.skip_connection = (in_channel == mlp[-1])
        widths = [in_channel] + list(mlp)
        for cin, cout in zip(widths[:-1], widths[1:]):
            self.mlp_convs.append(nn.Conv1d(cin, cout, 1))
            self.mlp_bns.append(nn.BatchNorm1d(cout))
        self.boundary_aware = nn.Sequential(
            nn.Conv1d(3, 16, 1),
            nn.BatchNorm1d(16),
            nn.ReLU(),
            nn.Conv1d(16, mlp[-1], 1))

    def forward(self, xyz1, xyz2, points1, points2):
        B, N, _ = xyz1.shape
        x, perm = _propagate_rows(xyz1, xyz2, points1, points2, 4)
        att = self.attention                                        # :279-280

        def trunk():
            a = rowmlp.conv_bn_act(att[0], att[1], x, rowmlp.ACT_RELU, perm=perm)
            g = rowmlp.gate_rows(x, rowmlp.conv_rows(att[3], a, out_gap=-perm))
            y = rowmlp.mlp_rows(self.mlp_convs, self.mlp_bns, g, perm=perm)
            if self.skip_connection:
                y = y + rowmlp.ungap_rows(g, perm)                  # :292-293
            return y

        # the boundary term depends on the coordinates only: all of it but its last conv runs on its own stream beside
        # the trunk; that conv (no BatchNorm behind it) adds the trunk's rows in its epilogue -- x = trunk + edge (:296)
        # without an addition pass
        ba = list(self.boundary_aware)
        if isinstance(ba[-1], (nn.Conv1d, nn.Conv2d)) and len(ba) >= 2:
            y, h = run_branches(xyz1.device, [trunk, lambda: _seq_rows(ba[:-1], xyz1.reshape(B * N, 3))],
                                inputs=(xyz1,))                      # :283
            x = rowmlp.conv_rows(ba[-1], h, add=y)
        else:
            y, edge = run_branches(xyz1.device, [trunk, lambda: _seq_rows(ba, xyz1.reshape(B * N, 3))], inputs=(xyz1,))
            x = y + edge
        return x.view(B, N, -1).transpose(1, 2)
