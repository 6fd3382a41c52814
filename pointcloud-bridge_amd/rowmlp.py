"""Shared pointwise MLP on channels-last rows: Conv(1x1) -> BatchNorm -> activation [-> max over
the neighbour axis], the block the reference repeats in every SetAbstraction / FeaturePropagation /
EdgeConv layer (models/pointnet2_utils.py:149-154, :207-209, :353-356; models/DGCNN.py:134-148).

ONE engine, two row types, selected with `set_precision`:

  "fp32"  (default)  fp32 rows on the exact fp32 matrix cores (csrc/gemm_f32.hip).  The parity mode:
                     network logits stay within 1e-4 relative of the reference
                     (tests/test_gpu_modules.py).  No library GEMM, no ATen BatchNorm.
  "bf16"             bf16 rows / fp32 statistics and master weights (csrc/gemm.hip).  The throughput
                     mode bench.py runs (BASELINE.json config 2 names bf16).

In both, a layer keeps only its pre-BatchNorm GEMM output; the BatchNorm / activation algebra of the
neighbouring layers is applied while the next GEMM loads its operand, batch statistics and the
BatchNorm-backward sums come out of GEMM epilogues, and a whole stack is enqueued by one native call
per direction (csrc/stack.hip).  The host code below is the same for both modes -- only the dtype
and the padding quantum (16 bytes = 8 bf16 or 4 fp32 columns) differ.

The parameters live in the caller's stock nn.Conv*/nn.BatchNorm* modules and their running
statistics are kept exactly as nn.BatchNorm does, so state_dicts stay interchangeable.
nn.SyncBatchNorm layers all-reduce their statistics over the process group (parallel.py).
"""
import ctypes
import os
import threading
import weakref

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import ops as _ops
from .ops import _launch, apply_concurrency_hint, on_device

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


class _Mode:
    """Row type of the engine: torch dtype, columns per 16-byte chunk, entry-point suffix, pcb_dtype."""

    def __init__(self, dtype, quantum, suffix, code):
        self.dtype, self.q, self.sfx, self.code = dtype, quantum, suffix, code

    def pad(self, k):
        return (k + self.q - 1) // self.q * self.q


_MODES = {"bf16": _Mode(torch.bfloat16, 8, "bf16", 0), "fp32": _Mode(torch.float32, 4, "f32", 1)}

# The row type in force is SCOPED, not global: a stack of `precision(...)` contexts per thread on top of a process-wide
# default.  A module (a whole network, one set-abstraction level) carries its own precision through
# bind_precision(module, mode): its forward pass then runs inside that context, whatever surrounds it, so two
# networks of different precision interleave in one process and threads do not see each other's choice.  The
# backward pass of every autograd function uses the row type its forward pass ran in (saved per call).
_DEFAULT_PRECISION = "fp32"
_tls = threading.local()


def _check_precision(mode):
    if mode not in _MODES:
        raise ValueError(f"precision must be 'fp32' or 'bf16', got {mode!r}")
    return mode


def set_precision(mode):
    """Process-wide DEFAULT row type: 'fp32' (parity mode) or 'bf16' (throughput mode).  Scopes opened with
    `precision(...)` and modules bound with bind_precision() take precedence."""
    global _DEFAULT_PRECISION
    _DEFAULT_PRECISION = _check_precision(mode)


class precision:
    """Context manager: the row type of every engine call made inside it by this thread; restores on exit.
        with rowmlp.precision("bf16"):
            logits = model(xyz, colors)"""

    def __init__(self, mode):
        self.mode = _check_precision(mode)

    def __enter__(self):
        stack = getattr(_tls, "stack", None)
        if stack is None:
            stack = _tls.stack = []
        stack.append(self.mode)
        return self

    def __exit__(self, *exc):
        _tls.stack.pop()
        return False


def bind_precision(module, mode):
    """Make `mode` ('fp32' / 'bf16', None removes the binding) a property of `module`: its forward pass (and everything it
    calls) runs in that row type regardless of the surrounding scope or default.  Returns the module."""
    for h in getattr(module, "_pcb_precision_hooks", ()):
        h.remove()
    module._pcb_precision_hooks = ()
    module.pcb_precision = None
    if mode is None:
        return module
    module.pcb_precision = _check_precision(mode)

    def enter(mod, args):
        stack = getattr(_tls, "stack", None)
        if stack is None:
            stack = _tls.stack = []
        stack.append(mod.pcb_precision)

    def leave(mod, args, out):
        _tls.stack.pop()

    module._pcb_precision_hooks = (module.register_forward_pre_hook(enter),
                                   module.register_forward_hook(leave, always_call=True))
    return module


def get_precision():
    stack = getattr(_tls, "stack", None)
    return stack[-1] if stack else _DEFAULT_PRECISION


def is_bf16():
    return get_precision() == "bf16"


def mode():
    return _MODES[get_precision()]


def pad_cols(k):
    """k rounded up to the current row type's 16-byte chunk (8 bf16 / 4 fp32 columns)."""
    return mode().pad(k)


def pad8(k):
    return (k + 7) // 8 * 8


def _act_torch(x, act):
    if act == ACT_RELU:
        return F.relu(x)
    if act == ACT_LEAKY:
        return F.leaky_relu(x, negative_slope=0.2)
    return x


def _weight2d(conv):
    return conv.weight.view(conv.out_channels, conv.in_channels)


def _bn_rows_fp32(bn, x):
    """ATen BatchNorm over rows [rows, C] with the module's exact running-stat bookkeeping -- only for
    layers the engine does not take (channel counts that are not multiples of 4: the 3..13-channel
    layers of the bridge encoders, models/attention_modules.py)."""
    if isinstance(bn, nn.SyncBatchNorm):
        return bn(x)  # statistics all-reduced over the process group (parallel.sync_batchnorm)
    eaf = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        eaf = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
    use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    return F.batch_norm(
        x,
        bn.running_mean if (not bn.training or bn.track_running_stats) else None,
        bn.running_var if (not bn.training or bn.track_running_stats) else None,
        bn.weight, bn.bias, use_batch, eaf, bn.eps)


def padded_weight_from(weight, kp, perm, quantum=8):
    """[Cout, kp] copy (bf16 for quantum 8, fp32 for 4) of a 1x1 conv weight ([Cout, K] or
    [Cout, K, 1(,1)]) in the column layout of the row buffer it multiplies -- the host statement of what
    pcb_prep_weights_* builds (tests compare the two):
      perm == 0  reference order, zero padded to kp
      perm = C > 0   grouped rows of pcb_group_rows_*: the C feature columns first, then the 3
                 centred coordinates (reference: coordinates first, :56/:347)
      perm = -D < 0  interpolate+concat rows of interpolate_concat: the first D (skip) columns stay
                 in place, the rest start at column pad(D)"""
    w = weight.reshape(weight.shape[0], -1)
    cout, k = w.shape
    wp = torch.zeros(cout, kp, dtype=torch.bfloat16 if quantum == 8 else torch.float32, device=w.device)
    if perm > 0:
        wp[:, :perm] = w[:, 3:3 + perm]
        wp[:, perm:perm + 3] = w[:, :3]
    elif perm < 0:
        d, dp = -perm, (-perm + quantum - 1) // quantum * quantum
        wp[:, :d] = w[:, :d]
        wp[:, dp:dp + k - d] = w[:, d:]
    else:
        wp[:, :k] = w
    return wp


def _counter(bn):
    """The module's num_batches_tracked if this forward has to increment it on the device (the
    momentum=None case increments it on the host in _bn_bookkeeping, which also reads it)."""
    if (bn.training and bn.track_running_stats and bn.num_batches_tracked is not None
            and bn.momentum is not None and bn.num_batches_tracked.is_cuda):
        return bn.num_batches_tracked
    return None


def _bn_bookkeeping(bn):
    """momentum to use for this call + the num_batches_tracked increment of nn.BatchNorm.forward."""
    eaf = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        eaf = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
    return eaf


def _rows(x, kp, m):
    """Make x [R,K] a contiguous [R,kp] operand of row type m (zero padded on the right)."""
    if (x.dtype == torch.float32 and x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] < kp
            and not (x.requires_grad and torch.is_grad_enabled())):
        # raw input columns (coordinates, colours): cast + pad in one launch instead of three
        out = torch.empty(x.shape[0], kp, dtype=m.dtype, device=x.device)
        with on_device(x.device):
            _launch("pcb_pad_rows_" + m.sfx, x.shape[0] * kp, x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], kp,
                    out.data_ptr())
        return out
    x = x.to(m.dtype)
    if x.shape[1] == kp:
        return x.contiguous()
    return F.pad(x, (0, kp - x.shape[1]))


# ---------------------------------------------------------------------------------------------
# Inference with constant weights: the prepared operands and the eval-mode BatchNorm constants of a
# stack depend on parameters and running statistics only, so a no-grad eval-mode call keeps them and
# the next call skips their preparation (23 + 34 launches of a PN2-MSG forward pass).
# Validation: every tensor's version counter (optimizer steps through torch.optim, load_state_dict,
# in-place edits), the first tensor's identity and address, AND a process-wide generation counter for
# the updates that bypass version counters: parallel.FlatAdam steps one flat buffer the parameters
# are views of, and the library updates running statistics through raw pointers in every
# training-mode forward.  Both call note_parameter_update().
# ---------------------------------------------------------------------------------------------
_eval_operands = {}
_generation = 0


def note_parameter_update(weights=True):
    """Invalidate the eval-mode operand cache: parameters or running statistics were updated by a path
    that does not bump tensor version counters (a flat-buffer optimizer step, a library-side
    running-statistics update).  weights=False: only running statistics changed (every training-mode
    forward says so) -- the per-step operand table (prepare_step) stays valid."""
    global _generation, _weights_generation
    _generation += 1
    if weights:
        _weights_generation += 1


# ---------------------------------------------------------------------------------------------
# Training: the GEMM operands of EVERY stack prepared by ONE launch per optimiser step.
# A stack's forward pass used to open with its own small operand-preparation launch (~20 per PN2-MSG
# step, each a dependent 5 us launch in front of the first GEMM).  The weights only change in the
# optimiser step, so the trainer calls prepare_step() right after it: one kernel walks a table (in
# device memory, rebuilt only when the set of stacks changes) of every registered stack's layers and
# fills their persistent operand buffers; the forward passes that follow find their entry marked
# prepared and tell the library to skip the launch (need_wt0 bit 2).  A stack registers itself the
# first time it runs; an entry is valid for exactly the parameter tensors (storage addresses, version
# counters, weight generation) it was prepared from -- anything else falls back to the per-stack launch.
#
# OWNERSHIP (round 3).  The registry is an object, StepOperands: one per model that asks for one
# (attach_step_operands(model): the model's forward pass registers its stacks there, prepare_step(model) fills them),
# plus a process default for callers that never attach.  So a captured hipGraph of model A bakes in the table and the
# buffers of A's OWN set: registering, training or dropping model B cannot touch them.  Inside a set nothing that a
# launch may have seen is ever freed: a table or an operand buffer that is replaced (a new stack registered, a
# parameter moved to other storage) is RETIRED -- kept alive by the set -- because a captured graph may still hold its
# address.  Tables are kept per (device, row type).
# ---------------------------------------------------------------------------------------------
_weights_generation = 0
_step_enabled = True


def set_step_operands(flag):
    """False: every stack call prepares its own operands in a buffer of its own, as before prepare_step existed
    (A/B runs, tests).  Returns the previous setting."""
    global _step_enabled
    old, _step_enabled = _step_enabled, bool(flag)
    return old


class _StepEntry:
    __slots__ = ("refs", "ptrs", "rows", "wbuf", "serial", "versions", "code")

    def alive(self):
        return all(r() is not None for r in self.refs)


def _step_rows(weights, Kp, perm, need_wt0, q):
    """(C, k, kp, perm, wp offset, wt offset or -1) per layer with a weight -- the layout pcb_mlp_stack_* uses
    inside wbuf (csrc/stack.hip: parse).  weights[0] is None for a gathered first layer (Kp = its width)."""
    rows, off, kp = [], 0, Kp
    for l, w in enumerate(weights):
        if w is None:
            continue
        C = w.shape[0]
        k = w.numel() // C
        wp = off
        off += C * kp
        wt = -1
        if l > 0 or need_wt0:
            wt = off
            off += C * kp
        rows.append((l, C, k, kp, perm if l == 0 else 0, wp, wt))
        kp = C
    return rows, off


class StepOperands:
    """The persistent GEMM operands of a set of stacks (normally: one model's) and the device tables one launch per
    optimiser step fills them from.  See the section comment above for ownership and lifetime."""

    def __init__(self):
        self.entries = {}
        self.tables = None        # {(device, mode code): (int64 table tensor, rows, workgroups)} or None when stale
        self.prepared = (-1, -1)  # (serial, _weights_generation) of the last prepare()
        self.serial = 0
        self.stats = [0, 0]       # stack calls that found their operands prepared / that prepared them themselves
        self.retired = []         # replaced tables / operand buffers: a captured graph may still hold their addresses

    def lookup(self, kind, weights, Kp, perm, need_wt0, m, nelem, dev):
        """(wbuf, flag) for a training-mode stack call: the entry's persistent operand buffer and 4 if the last
        prepare() filled it from exactly these parameters, else 0 (the call prepares them itself)."""
        first = next(w for w in weights if w is not None)
        key = (kind, id(first), Kp, perm, len(weights), m.code, int(need_wt0))
        ptrs = tuple(0 if w is None else w.data_ptr() for w in weights)
        e = self.entries.get(key)
        if e is None or e.ptrs != ptrs or not e.alive() or e.wbuf.numel() != max(nelem, 1) or e.wbuf.device != dev:
            if e is not None:
                self.retired.append(e.wbuf)
            e = _StepEntry()
            e.refs = [weakref.ref(w) for w in weights if w is not None]
            e.ptrs = ptrs
            e.rows, _ = _step_rows(weights, Kp, perm, need_wt0, m.q)
            e.wbuf = torch.empty(max(nelem, 1), dtype=m.dtype, device=dev)
            e.serial = -1
            e.versions = None
            e.code = m.code
            self.entries[key] = e
            self._retire_tables()
            return e.wbuf, 0
        if (e.serial == self.prepared[0] and self.prepared[1] == _weights_generation
                and e.versions == tuple(w._version for w in weights if w is not None)):
            self.stats[0] += 1
            return e.wbuf, 4
        self.stats[1] += 1
        return e.wbuf, 0

    def _retire_tables(self):
        if self.tables is not None:
            self.retired.extend(t for t, _, _ in self.tables.values())
            self.tables = None

    def prepare(self):
        """One launch per (device, row type): the operands of every registered stack from the current weights."""
        dead = [k for k, e in self.entries.items() if not e.alive()]
        for k in dead:
            self.retired.append(self.entries.pop(k).wbuf)
            self._retire_tables()
        if not self.entries:
            return
        if self.tables is None:
            rows = {}
            for e in self.entries.values():
                live = [r() for r in e.refs]
                base, es = e.wbuf.data_ptr(), e.wbuf.element_size()
                it = iter(live)
                vals, most = rows.setdefault((e.wbuf.device, e.code), ([], [0]))   # most[0]: workgroups so far (one 32 x 32 tile each)
                for (l, C, k, kp, perm, wp, wt) in e.rows:
                    w = next(it)
                    vals += [w.data_ptr(), base + es * wp, 0 if wt < 0 else base + es * wt, C, k, kp, perm, most[0]]
                    most[0] += ((C + 31) // 32) * ((kp + 31) // 32)
            self.tables = {key: (torch.tensor(vals, dtype=torch.int64, device=key[0]), len(vals) // 8, most[0])
                           for key, (vals, most) in rows.items()}
        self.serial += 1
        for (dev, code), (table, n, most) in self.tables.items():
            sfx = "bf16" if code == _MODES["bf16"].code else "f32"
            with on_device(dev):
                _launch("pcb_prep_weights_table_" + sfx, n, table.data_ptr(), n, most)
        for e in self.entries.values():
            e.serial = self.serial
            e.versions = tuple(r()._version for r in e.refs)
        self.prepared = (self.serial, _weights_generation)


_default_operands = StepOperands()
_step_stats = _default_operands.stats   # (tests read the default set's counters)


def _current_operands():
    stack = getattr(_tls, "opsets", None)
    return stack[-1] if stack else _default_operands


def attach_step_operands(module):
    """Give `module` (a whole network) its OWN StepOperands: stacks that run inside its forward pass register there,
    prepare_step(module) fills them.  Returns the set (idempotent)."""
    ops = getattr(module, "_pcb_step_operands", None)
    if ops is not None:
        return ops
    ops = module._pcb_step_operands = StepOperands()

    def enter(mod, args):
        stack = getattr(_tls, "opsets", None)
        if stack is None:
            stack = _tls.opsets = []
        stack.append(mod._pcb_step_operands)

    def leave(mod, args, out):
        _tls.opsets.pop()

    module.register_forward_pre_hook(enter)
    module.register_forward_hook(leave, always_call=True)
    return ops


def _step_operands(kind, weights, Kp, perm, need_wt0, m, nelem, dev):
    if not _step_enabled:
        return torch.empty(max(nelem, 1), dtype=m.dtype, device=dev), 0
    return _current_operands().lookup(kind, weights, Kp, perm, need_wt0, m, nelem, dev)


def prepare_step(module=None):
    """Call after every optimiser step (the trainer and bench.py do): prepares the operands of all registered
    stacks -- of `module`'s own set (attach_step_operands) or, without an argument, of the process default set --
    from the current weights with one launch per row type.  Purely an optimisation: stacks whose entry is
    not (or no longer) valid prepare their own operands as before.  An entry is trusted until a version counter
    of its weights moves or note_parameter_update() is called: code that edits weights through `.data` between
    this call and the forward pass must call one of the two (parallel.FlatAdam does)."""
    ops = _default_operands if module is None else getattr(module, "_pcb_step_operands", None)
    if ops is None:
        raise ValueError("prepare_step(module): call attach_step_operands(module) first")
    ops.prepare()


_eval_retired = []   # buffers of replaced / evicted cache entries that a captured graph has seen (see _eval_store)


def _eval_lookup(layers, first, extra):
    """(key, versions, cached buffers or None) for a no-grad call whose layers are all in eval mode."""
    versions = tuple([t._version for lay in layers for t in lay[:6] if t is not None]) + (first.data_ptr(), _generation)
    key = (id(first), _centring) + extra
    hit = _eval_operands.get(key)
    if hit is not None and hit[0]() is first and hit[1] == versions:
        if not hit[3][0] and torch.cuda.is_current_stream_capturing():
            hit[3][0] = True   # a hipGraph now holds these addresses
        return key, versions, hit[2]
    return key, versions, None


def _eval_store(key, first, versions, bufs):
    """Keep the prepared operands of an eval-mode call.  An entry that is replaced (the parameters moved on) or evicted
    is simply dropped -- unless a captured graph has used its buffers: such a graph is stale in its VALUES once the
    weights change (inference with constant weights is what it captured), but its replay must not walk freed memory,
    so those buffers are retired instead."""
    def drop(entry):
        if entry[3][0]:
            _eval_retired.append(entry[2])

    if len(_eval_operands) > 4096:
        for k in [k for k, e in _eval_operands.items() if e[0]() is None]:
            drop(_eval_operands.pop(k))
        if len(_eval_operands) > 4096:
            for e in _eval_operands.values():
                drop(e)
            _eval_operands.clear()
    old = _eval_operands.get(key)
    if old is not None:
        drop(old)
    _eval_operands[key] = (weakref.ref(first), versions, bufs, [torch.cuda.is_current_stream_capturing()])


# ---------------------------------------------------------------------------------------------
# SyncBatchNorm: the library hands the local statistics totals (2C floats per layer) to this callback
# between launches; it all-reduces them over the layer's process group in stream order.
# ---------------------------------------------------------------------------------------------
_SYNC_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)


class _SyncStruct(ctypes.Structure):
    _fields_ = [("allreduce", _SYNC_FN), ("ctx", ctypes.c_void_p), ("global_rows", ctypes.c_long)]


class _Sync:
    """pcb_sync for one stack call: owns the ctypes callback and finds the torch tensor behind the
    device pointer the library passes (it always lies inside one of the call's float buffers)."""

    def __init__(self, group, rows, buffers):
        self.group = group
        self.buffers = buffers
        self.error = None
        self._cb = _SYNC_FN(self._allreduce)
        world = dist.get_world_size(group)
        self.struct = _SyncStruct(self._cb, None, int(rows) * world)  # every rank holds the same number of rows

    def _allreduce(self, ptr, n, _ctx):
        try:
            for t in self.buffers:
                base = t.data_ptr()
                if base <= ptr < base + t.numel() * 4:
                    off = (ptr - base) // 4
                    dist.all_reduce(t.view(-1)[off:off + n], op=dist.ReduceOp.SUM, group=self.group)
                    return 0
            raise RuntimeError("SyncBatchNorm: statistics pointer outside the stack's buffers")
        except BaseException as e:  # an exception must not unwind through the C frames
            self.error = e
            return -1

    def ref(self):
        return ctypes.byref(self.struct)

    def check(self):
        if self.error is not None:
            err, self.error = self.error, None
            raise err


def _sync_group(bns):
    """The process group to synchronise over if these layers are training-mode SyncBatchNorm under an
    initialised process group of more than one rank; else None.  (False, None) = no sync."""
    sync = [bn for bn in bns if isinstance(bn, nn.SyncBatchNorm)]
    if not sync or not any(bn.training for bn in sync):
        return False, None
    if not dist.is_available() or not dist.is_initialized():
        return False, None
    group = getattr(sync[0], "process_group", None)
    if dist.get_world_size(group) == 1:
        return False, None
    if len(sync) != len(bns):
        raise NotImplementedError("a stack mixes SyncBatchNorm and BatchNorm layers")
    return True, group


# ---------------------------------------------------------------------------------------------
# bf16 rows: pre-BatchNorm outputs stored CENTRED (include/pcb_hip.h: pcb_gemm_nt_stats_bf16).
# A bf16 value carries an absolute error of 2^-9 |y| and train-mode BatchNorm divides by std(y), so a layer stored as
# is loses 2^-9 (|mean|/std + 1) of the normalised signal; BatchNorm does not see a per-channel constant in front of
# it (the conv bias cancels the same way, models/pointnet2_utils.py:149-151), so the GEMM epilogue subtracts one close
# to the batch mean before rounding.  The constant is the PREVIOUS step's batch mean of the layer, kept in a plain
# fp32 tensor on the BatchNorm module (`_pcb_centre`: not a registered buffer -- state_dict keys stay the reference's)
# and moved by the layer's finalize kernel whenever the batch mean has drifted more than std/4 away from it; the first
# training call of a layer runs its GEMM twice (probe: learn the mean, then store centred on it).  Eval mode centres on running_mean - bias exactly.
# ---------------------------------------------------------------------------------------------
CF_PROBE, CF_EVAL = 1, 2
_centring = os.environ.get("PCB_CENTRE", "1") != "0"


def set_centring(flag):
    """Enable / disable centred storage of pre-BatchNorm rows in bf16 mode (A/B runs, tests).  Returns the previous setting."""
    global _centring
    old, _centring = _centring, bool(flag)
    return old


def _centre_args(bn, C, training, sync=False):
    """(centre tensor or None, flags) for one layer of a bf16 stack call (desc slots [16], [17])."""
    if not _centring:
        return None, 0
    if not training:
        return None, CF_EVAL
    dev = bn.weight.device if bn.weight is not None else (bn.running_mean.device if bn.running_mean is not None else None)
    if dev is None or dev.type != "cuda":
        return None, 0
    c = getattr(bn, "_pcb_centre", None)
    if c is None or c.device != dev or c.numel() != C:
        if torch.cuda.is_current_stream_capturing():
            return None, 0   # a buffer that outlives the capture cannot be created inside it: this call stays uncentred
        c = torch.zeros(C, dtype=torch.float32, device=dev)
        bn._pcb_centre = c
        bn._pcb_centre_ok = False
    flags = 0
    if not getattr(bn, "_pcb_centre_ok", False):
        # (under SyncBatchNorm the probe's statistics are all-reduced like any other: every rank learns the GLOBAL mean)
        flags = CF_PROBE
        bn._pcb_centre_ok = True
    return c, flags


def centre_state(module):
    """{submodule name: centre tensor (a copy)} of every BatchNorm layer under `module` that holds an estimate.  The centres
    are engine state like the running statistics but NOT part of state_dict() (the reference's keys are kept): two models
    with equal parameters and equal centres produce equal bits; with different centres, equal values up to bf16 rounding."""
    return {name: m._pcb_centre.clone() for name, m in module.named_modules()
            if getattr(m, "_pcb_centre", None) is not None and getattr(m, "_pcb_centre_ok", False)}


def load_centre_state(module, state):
    """Install centres saved by centre_state() (layers without an entry forget theirs and probe again)."""
    for name, m in module.named_modules():
        if name in state:
            m._pcb_centre = state[name].clone()
            m._pcb_centre_ok = True
        elif hasattr(m, "_pcb_centre"):
            del m._pcb_centre
            m._pcb_centre_ok = False


def reset_centres(module):
    """Forget every centre under `module` (the next training call of each layer probes again): what a caller does
    beside reset_running_stats() / load_state_dict() when it wants a replay to reproduce an earlier run bit for bit."""
    load_centre_state(module, {})


_MAX_PARTS = 768     # slabs a gemm_nt launch writes at most (PCB_MAX_SLABS)
_GATHER_PARTS = 1024  # a gathered first layer


class _FusedStack(torch.autograd.Function):
    """A stack of L layers act(BN(. W^T)) on rows with everything between the GEMMs fused
    (csrc/gemm.hip, csrc/gemm_f32.hip): layer l's BatchNorm+activation is applied while layer l+1
    loads its operand, batch statistics come out of the GEMM epilogue, the backward recomputes dy on
    load.  Per layer only y_l = x_l W_l^T is stored.  The last layer's activation is materialised
    (rows) or max-pooled over `pool` consecutive rows.

    Flat argument list: x, act, pool, perm, stat_repeat, L, mode, sync group (or False), then per layer
    (weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps,
    num_batches_tracked or None -- incremented by the finalize kernel --, centre or None, centre flags: see
    _centre_args); optionally, behind the layers,
    (add1, add2 or None, sh1, sh2): fp32 rows [R >> sh, C_0] added to layer 0's product, each standing for 2^sh
    consecutive rows (conv_bn_act_levels; pcb_hip.h desc slot [15]).  Layer 0's weight may then be a column slice of
    a wider weight."""

    NPER = 12
    NHEAD = 8

    @staticmethod
    def forward(ctx, x, act, pool, perm, stat_repeat, L, m, group, *flat):
        dev = x.device
        R, Kp = x.shape
        layers = [flat[i * _FusedStack.NPER:(i + 1) * _FusedStack.NPER] for i in range(L)]
        adds = flat[L * _FusedStack.NPER:] or None
        need_dx = bool(x.requires_grad)
        widths = [t[0].shape[0] for t in layers]
        lib = _lib.load()
        # caller-owned buffers of pcb_mlp_stack_forward/backward (include/pcb_hip.h)
        ybuf = torch.empty(R * sum(widths), dtype=m.dtype, device=dev)
        stz = torch.empty(10 * sum(widths), dtype=torch.float32, device=dev)
        parts = torch.empty(_MAX_PARTS * 2 * max(widths), dtype=torch.float32, device=dev)
        ext = None
        if adds is not None:
            add1, add2, sh1, sh2 = adds
            for a, sh in ((add1, sh1), (add2, sh2)):
                if a is not None and not (a.dtype == torch.float32 and a.is_contiguous()
                                          and a.shape == (R >> sh, widths[0]) and (R >> sh) << sh == R):
                    raise ValueError("repeated addends: fp32 rows [R >> sh, C] with R a multiple of 2^sh")
            w0 = layers[0][0]
            ext = (ctypes.c_longlong * 7)(add1.data_ptr(), sh1, 0 if add2 is None else add2.data_ptr(), sh2, 0, 0,
                                          0 if w0.is_contiguous() else w0.stride(0))
        strided = not layers[0][0].is_contiguous()
        desc = _stack_desc(layers, widths, ybuf, R, m, ext=ext)
        ready = 0
        cache_key = None
        training = any(t[6] for t in layers)
        # pipelined inference: the next batch's FPS may be running beside this pass; training: a hint left
        # over from the backward pass is dropped once its event has completed
        apply_concurrency_hint()
        if not torch.is_grad_enabled() and not training and not strided:
            cache_key, versions, hit = _eval_lookup(layers, layers[0][0], ("stack", Kp, perm, L, m.code))
            if hit is not None:
                wbuf, stz = hit
                ready = 2
        if training:
            note_parameter_update(weights=False)  # running statistics change under the eval cache's feet
        if not ready:
            nw = lib.pcb_mlp_stack_wbuf_elems(L, desc, Kp, int(need_dx))
            if strided:
                # a column slice of a wider weight is a fresh view object per call: no registry entry, the call
                # prepares its operands itself (one small launch)
                wbuf = torch.empty(nw, dtype=m.dtype, device=dev)
            elif cache_key is None:
                # the stack's persistent operand buffer; flag 4: already prepared from these weights (prepare_step)
                wbuf, ready = _step_operands("stack", [t[0] for t in layers], Kp, perm, need_dx, m, nw, dev)
            else:
                wbuf = torch.empty(nw, dtype=m.dtype, device=dev)
        fdesc = (ctypes.c_double * (2 * L))(*[float(v) for t in layers for v in (t[7], t[8])])
        C = widths[-1]
        if pool:
            out = torch.empty(R // pool, C, dtype=m.dtype, device=dev)
            arg = torch.empty(R // pool, C, dtype=torch.uint8, device=dev)
        else:
            out = torch.empty(R, C, dtype=m.dtype, device=dev)
            arg = None
        sync = _Sync(group, R, (stz, parts)) if group is not False else None
        with on_device(dev):
            _launch("pcb_mlp_stack_forward", 0, m.code, L, desc, fdesc, x.data_ptr(), R, Kp, perm, act, pool,
                    int(need_dx) | ready, int(stat_repeat), 0, wbuf.data_ptr(), stz.data_ptr(), parts.data_ptr(),
                    _MAX_PARTS, None if sync is None else sync.ref(), out.data_ptr(),
                    0 if arg is None else arg.data_ptr(), check=(None if sync is None else sync.check))
        if cache_key is not None and not ready:
            _eval_store(cache_key, layers[0][0], versions, (wbuf, stz))
        ctx.save_for_backward(x, arg, ybuf, stz, wbuf, *[t[0] for t in layers])
        ctx.cfg = (act, pool, perm, L, need_dx, m, group,
                   [(t[1] is not None, t[2] is not None, bool(t[6])) for t in layers])
        ctx.adds = None if adds is None else (adds[0].shape, None if adds[1] is None else adds[1].shape, adds[2], adds[3])
        return out

    @staticmethod
    def backward(ctx, g):
        act, pool, perm, L, need_dx, m, group, flags = ctx.cfg
        saved = ctx.saved_tensors
        x, arg, ybuf, stz, wbuf = saved[:5]
        weights = saved[5:5 + L]
        widths = [w.shape[0] for w in weights]
        dev = x.device
        R, Kp = x.shape
        lib = _lib.load()
        g = g.float().contiguous() if pool else g.to(m.dtype).contiguous()
        H = _FusedStack.NHEAD
        grads = [None] * (L * _FusedStack.NPER)
        outs = []
        kp, ws_elems = Kp, 0
        for l, w in enumerate(weights):
            C = widths[l]
            has_bias, has_affine, _ = flags[l]
            base = l * _FusedStack.NPER
            dw = grads[base + 0] = (torch.empty(w.shape, dtype=w.dtype, device=dev)
                                    if ctx.needs_input_grad[H + base] else None)
            dbias = grads[base + 1] = torch.empty(C, dtype=torch.float32, device=dev) if has_bias else None
            dgamma = grads[base + 2] = torch.empty(C, dtype=torch.float32, device=dev) if has_affine else None
            dbeta = grads[base + 3] = torch.empty(C, dtype=torch.float32, device=dev) if has_affine else None
            outs.append((dw, dgamma, dbeta, dbias))
            if dw is not None:
                ws_elems += lib.pcb_gemm_tn_workspace(R, C, kp)  # one slab region per layer
            kp = C
        layers = [(w, None, None, None, None, None, flags[l][2]) for l, w in enumerate(weights)]
        ext, dadds = None, ()
        if ctx.adds is not None:
            s1, s2, sh1, sh2 = ctx.adds
            base = H + L * _FusedStack.NPER
            d1 = torch.empty(s1, dtype=torch.float32, device=dev) if ctx.needs_input_grad[base] else None
            d2 = torch.empty(s2, dtype=torch.float32, device=dev) if (s2 is not None and ctx.needs_input_grad[base + 1]) else None
            if d1 is None and d2 is not None:   # the kernel writes the finer level's sums on its way to the coarser one's
                d1 = torch.empty(s1, dtype=torch.float32, device=dev)
            ext = (ctypes.c_longlong * 7)(0, sh1, 0, sh2, 0 if d1 is None else d1.data_ptr(),
                                          0 if d2 is None else d2.data_ptr(),
                                          0 if weights[0].is_contiguous() else weights[0].stride(0))
            dadds = (d1 if ctx.needs_input_grad[base] else None, d2, None, None)
        desc = _stack_desc(layers, widths, ybuf, R, m, outs, ext=ext)
        want_dx = bool(ctx.needs_input_grad[0]) and need_dx
        dx = torch.empty(R, Kp, dtype=m.dtype, device=dev) if want_dx else None
        # the two gradient ping-pong slots (and, for a wide top layer, its dy written out once): sized by the library
        nz = lib.pcb_mlp_stack_dzbuf_elems(m.code, L, desc, R, Kp, pool, 0)
        dzbuf = torch.empty(nz, dtype=m.dtype, device=dev) if nz else None
        parts = torch.empty(_MAX_PARTS * 2 * max(widths + [Kp]), dtype=torch.float32, device=dev)
        ws = torch.empty(max(ws_elems, 1), dtype=torch.float32, device=dev)
        sync = _Sync(group, R, (stz, parts)) if group is not False else None
        apply_concurrency_hint()
        with on_device(dev):
            _launch("pcb_mlp_stack_backward", 0, m.code, L, desc, x.data_ptr(), g.data_ptr(),
                    0 if arg is None else arg.data_ptr(), R, Kp, perm, act, pool, int(need_dx), 0, wbuf.data_ptr(),
                    stz.data_ptr(), parts.data_ptr(), _MAX_PARTS, None if sync is None else sync.ref(), ws.data_ptr(),
                    0 if dzbuf is None else dzbuf.data_ptr(), 0 if dx is None else dx.data_ptr(),
                    check=(None if sync is None else sync.check))
        return (dx, None, None, None, None, None, None, None, *grads, *dadds)


class _PointLinear(torch.autograd.Function):
    """out [n, C] fp32 = x [n, K] (bf16) @ w [C, K]^T (fp32 master weight, bf16 operand): the small
    per-point GEMM in front of a gathered stack, with an UNROUNDED result (hipBLASLt's fp32 kernels
    run these skinny shapes at a fraction of the bandwidth; a bf16 result would lose the difference
    of two nearby points).  Backward: dW by the split-row MFMA kernel, dx as bf16 rows."""

    @staticmethod
    def forward(ctx, x, w):
        n, K = x.shape
        C = w.shape[0]
        dev = x.device
        wp = torch.empty(C, K, dtype=torch.bfloat16, device=dev)
        wt = torch.empty(K, C, dtype=torch.bfloat16, device=dev) if x.requires_grad else None
        out = torch.empty(n, C, dtype=torch.float32, device=dev)
        if w.dtype != torch.float32 or w.stride(1) != 1 or w.shape[1] != K or w.stride(0) < K:
            raise TypeError("point_linear expects an fp32 weight [C, K] with contiguous rows (a column slice of a wider weight is fine)")
        desc = (ctypes.c_longlong * 8)(w.data_ptr(), wp.data_ptr(), 0 if wt is None else wt.data_ptr(), C, K, K, 0, w.stride(0))
        with on_device(dev):
            _launch("pcb_prep_weights_bf16", C * K, 1, desc)
            _launch("pcb_gemm_nt_f32out_bf16", n * (K + 2 * C), x.data_ptr(), wp.data_ptr(), n, C, K, out.data_ptr())
        ctx.save_for_backward(x, wt)
        ctx.wshape = w.shape
        return out

    @staticmethod
    def backward(ctx, g):
        x, wt = ctx.saved_tensors
        n, K = x.shape
        C = ctx.wshape[0]
        dev = x.device
        gb = g.to(torch.bfloat16).contiguous()
        lib = _lib.load()
        dw = dx = None
        with on_device(dev):
            if ctx.needs_input_grad[1]:
                dw = torch.empty(ctx.wshape, dtype=torch.float32, device=dev)
                ws = torch.empty(lib.pcb_gemm_tn_workspace(n, C, K), dtype=torch.float32, device=dev)
                _launch("pcb_gemm_tn_bf16", 2 * n * (C + K), 0, gb.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, x.data_ptr(),
                        0, 0, 0, n, C, K, ws.data_ptr(), dw.data_ptr(), 0, 0)
            if ctx.needs_input_grad[0] and wt is not None:
                dx = torch.empty(n, K, dtype=torch.bfloat16, device=dev)
                _launch("pcb_gemm_nt_bf16", 2 * n * (C + K), 0, gb.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 0, 0, wt.data_ptr(), n, K,
                        C, dx.data_ptr(), 0, 0)
        return dx, dw


class _SplitCols(torch.autograd.Function):
    """w[:, :d] and w[:, d:] (views of w, rows w.shape[1] floats apart).  Backward is ONE concatenation: a pair of plain
    slices costs a zero fill, a copy and an accumulation per piece in the autograd engine (six launches
    for a weight the GEMMs read in two parts)."""

    @staticmethod
    def forward(ctx, w, d):
        ctx.d, ctx.shape = d, w.shape
        # views: the consumers (pcb_prep_weights_* through point_linear, the coordinate term of pcb_gather_add_bf16)
        # read a column slice in place through its row stride -- two copy launches per gathered stack otherwise
        return w[:, :d], w[:, d:]

    @staticmethod
    def backward(ctx, ga, gb):
        d, (c, k) = ctx.d, ctx.shape
        ref = ga if ga is not None else gb
        if ga is None:
            ga = ref.new_zeros(c, d)
        if gb is None:
            gb = ref.new_zeros(c, k - d)
        return torch.cat([ga, gb], dim=1), None


class _SplitColsN(torch.autograd.Function):
    """Column blocks of a 2-D weight as views (widths in order); backward is one concatenation (see _SplitCols)."""

    @staticmethod
    def forward(ctx, w, widths):
        ctx.widths, ctx.rows = widths, w.shape[0]
        outs, o = [], 0
        for d in widths:
            outs.append(w[:, o:o + d])
            o += d
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        return torch.cat([g if g is not None else ref.new_zeros(ctx.rows, d) for g, d in zip(gs, ctx.widths)], dim=1), None


def split_cols(w, d):
    """(w[:, :d], w[:, d:]) of a 2-D weight as views, with a single-launch backward (see _SplitCols)."""
    return _SplitCols.apply(w, int(d))


def point_linear(x, w):
    """x [n, K] @ w[C, K]^T -> fp32 [n, C] with bf16 operands (K, C multiples of 8); see _PointLinear."""
    # (w may be a column slice of a wider weight: the operand preparation reads it through its row stride)
    return _PointLinear.apply(x.to(torch.bfloat16).contiguous(), w if w.stride(-1) == 1 else w.contiguous())


class _GatheredStack(torch.autograd.Function):
    """A grouped stack whose FIRST layer is evaluated per point and gathered (csrc/gatherlin.hip):
    y0[(s,j)] = u[idx[s,j]] + v[s] + wx (xyz[idx[s,j]] - ctr[s]) with per-point products u [B*N,C0],
    v [B*S,C0] (fp32, from point_linear; v optional) and, optionally, the coordinate columns wx
    [C0,3] of the first conv applied to the fp32 difference exactly as the reference forms it; then
    BatchNorm + activation and the remaining layers / pooling exactly as _FusedStack (bf16 rows).
    Backward returns du, dv, dwx; autograd takes du, dv on through the caller's point_linear calls.

    Flat argument list: u, v, wx, xyz, ctr, idx, ns, act, pool, L, then per layer (weight, bias,
    gamma, beta, running_mean, running_var, training, momentum, eps, num_batches_tracked or None, centre or None,
    centre flags); layer 0's weight is None."""

    NPER = 12
    NHEAD = 10

    @staticmethod
    def forward(ctx, u, v, wx, xyz, ctr, idx, ns, act, pool, L, *flat):
        dev = u.device
        m = _MODES["bf16"]
        B, S = idx.shape[0], idx.shape[1]
        N = u.shape[0] // B
        R = B * S * ns
        layers = [flat[i * _GatheredStack.NPER:(i + 1) * _GatheredStack.NPER] for i in range(L)]
        widths = [u.shape[1]] + [t[0].shape[0] for t in layers[1:]]
        lib = _lib.load()
        ybuf = torch.empty(R * sum(widths), dtype=torch.bfloat16, device=dev)
        stz = torch.empty(10 * sum(widths), dtype=torch.float32, device=dev)
        parts = torch.empty(_GATHER_PARTS * 2 * max(widths), dtype=torch.float32, device=dev)
        desc = _stack_desc(layers, widths, ybuf, R, m)
        ready = 0
        cache_key = None
        training = any(t[6] for t in layers)
        if not torch.is_grad_enabled() and not training:
            first = next(t for lay in layers for t in lay[:6] if t is not None)
            cache_key, versions, hit = _eval_lookup(layers, first, ("gathered", L))
            if hit is not None:
                wbuf, stz = hit
                ready = 2
        if training:
            note_parameter_update(weights=False)
        if not ready:
            nw = lib.pcb_mlp_stack_wbuf_elems(L, desc, 0, 0)
            if L > 1 and cache_key is None:
                wbuf, ready = _step_operands("gathered", [None] + [t[0] for t in layers[1:]], widths[0], 0, False, m, nw, dev)
            else:
                wbuf = torch.empty(max(nw, 1), dtype=torch.bfloat16, device=dev)
        fdesc = (ctypes.c_double * (2 * L))(*[float(x) for t in layers for x in (t[7], t[8])])
        gather = (ctypes.c_longlong * 16)(
            u.data_ptr(), 0 if v is None else v.data_ptr(), idx.data_ptr(), B, N, S, ns,
            0 if wx is None else xyz.data_ptr(), 0 if wx is None else ctr.data_ptr(),
            0 if wx is None else wx.data_ptr(), 3 if wx is None else wx.stride(0), 0, 0, 0, 0, 0)
        C = widths[-1]
        if pool:
            out = torch.empty(R // pool, C, dtype=torch.bfloat16, device=dev)
            arg = torch.empty(R // pool, C, dtype=torch.uint8, device=dev)
        else:
            out = torch.empty(R, C, dtype=torch.bfloat16, device=dev)
            arg = None
        with on_device(dev):
            _launch("pcb_mlp_stack_forward", 0, m.code, L, desc, fdesc, 0, R, 0, 0, act, pool, ready, 1, gather,
                    wbuf.data_ptr(), stz.data_ptr(), parts.data_ptr(), _GATHER_PARTS, None, out.data_ptr(),
                    0 if arg is None else arg.data_ptr())
        if cache_key is not None and not ready:
            _eval_store(cache_key, first, versions, (wbuf, stz))
        # reproducible mode: the inverted index of the gather (source point -> grouped rows) for the backward pass
        # (grad mode is off inside a Function's forward: needs_input_grad says whether a backward pass can follow)
        det = _ops.det_index(idx, N) if (_ops.deterministic() and any(ctx.needs_input_grad)) else (None, None)
        ctx.save_for_backward(idx, arg, ybuf, stz, wbuf, xyz if wx is not None else None,
                              ctr if wx is not None else None, det[0], det[1], *[t[0] for t in layers[1:]])
        ctx.cfg = (act, pool, L, ns, B, N, S, widths, v is not None, wx is not None,
                   [(t[1] is not None, t[2] is not None, bool(t[6])) for t in layers])
        return out

    @staticmethod
    def backward(ctx, g):
        act, pool, L, ns, B, N, S, widths, has_v, has_wx, flags = ctx.cfg
        m = _MODES["bf16"]
        saved = ctx.saved_tensors
        idx, arg, ybuf, stz, wbuf, xyz, ctr, order, offsets = saved[:9]
        weights = [None] + list(saved[9:9 + L - 1])
        det = order is not None
        dev = idx.device
        R = B * S * ns
        lib = _lib.load()
        g = g.float().contiguous() if pool else g.to(torch.bfloat16).contiguous()
        H = _GatheredStack.NHEAD
        grads = [None] * (L * _GatheredStack.NPER)
        outs, ws_elems = [], 0
        for l in range(L):
            C = widths[l]
            has_bias, has_affine, _ = flags[l]
            base = l * _GatheredStack.NPER
            dw = None
            if l and ctx.needs_input_grad[H + base]:
                dw = grads[base + 0] = torch.empty_like(weights[l])
                ws_elems += lib.pcb_gemm_tn_workspace(R, C, widths[l - 1])  # one slab region per layer
            dbias = grads[base + 1] = torch.empty(C, dtype=torch.float32, device=dev) if has_bias else None
            dgamma = grads[base + 2] = torch.empty(C, dtype=torch.float32, device=dev) if has_affine else None
            dbeta = grads[base + 3] = torch.empty(C, dtype=torch.float32, device=dev) if has_affine else None
            outs.append((dw, dgamma, dbeta, dbias))
        layers = [(weights[l], None, None, None, None, None, flags[l][2]) for l in range(L)]
        desc = _stack_desc(layers, widths, ybuf, R, m, outs)
        du = torch.empty(B * N, widths[0], dtype=torch.float32, device=dev)   # zeroed / fully written by the library
        dv = torch.empty(B * S, widths[0], dtype=torch.float32, device=dev) if has_v else None
        slabs = lib.pcb_scatter_dy_slabs(B, S, widths[0], int(det))
        dwx = torch.empty(slabs, widths[0], 3, dtype=torch.float32, device=dev) if has_wx else None  # [0] = result
        gather = (ctypes.c_longlong * 16)(
            du.data_ptr(), 0 if dv is None else dv.data_ptr(), idx.data_ptr(), B, N, S, ns,
            0 if dwx is None else xyz.data_ptr(), 0 if dwx is None else ctr.data_ptr(),
            0 if dwx is None else dwx.data_ptr(), 3, int(det), order.data_ptr() if det else 0,
            offsets.data_ptr() if det else 0, 0, 0)
        nz = lib.pcb_mlp_stack_dzbuf_elems(_MODES["bf16"].code, L, desc, R, 0, pool, 1)
        dzbuf = torch.empty(nz, dtype=torch.bfloat16, device=dev) if nz else None
        parts = torch.empty(_MAX_PARTS * 2 * max(widths), dtype=torch.float32, device=dev)
        ws = torch.empty(max(ws_elems, 1), dtype=torch.float32, device=dev)
        apply_concurrency_hint()
        with on_device(dev):
            _launch("pcb_mlp_stack_backward", 0, m.code, L, desc, 0, g.data_ptr(), 0 if arg is None else arg.data_ptr(),
                    R, 0, 0, act, pool, 0, gather, wbuf.data_ptr(), stz.data_ptr(), parts.data_ptr(), _MAX_PARTS, None,
                    ws.data_ptr(), 0 if dzbuf is None else dzbuf.data_ptr(), 0)
        return (du, dv, None if dwx is None else dwx[0], None, None, None, None, None, None, None, *grads)


def _layer_args(conv, bn, with_weight=True, m=None, sync=False):
    """(weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, counter, centre, centre flags)
    of one layer; m: the row type of the call (centred storage is a bf16 matter)."""
    momentum = bn.momentum if bn.momentum is not None else _bn_bookkeeping(bn)
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    track = bn.track_running_stats and bn.running_mean is not None
    centre, cflags = (None, 0)
    if m is _MODES["bf16"]:
        centre, cflags = _centre_args(bn, conv.out_channels, training, sync)
    return [conv.weight if with_weight else None, conv.bias, bn.weight, bn.bias,
            bn.running_mean if (track or not training) else None,
            bn.running_var if (track or not training) else None,
            training, momentum, bn.eps, _counter(bn), centre, cflags]


def gathered_mlp(convs, bns, u, v, idx, act=ACT_RELU, pool=0, wx=None, xyz=None, ctr=None):
    """bf16 engine: the stack convs/bns applied to grouped rows whose first-layer products are
    given per point: u [B*N, C0] fp32 (source points), v [B*S, C0] fp32 (centroids) or None,
    idx [B,S,ns] int64; row (s,j) of layer 0's output is u[idx[s,j]] + v[s] (+ wx (xyz_j - ctr_s)
    with wx [C0,3] fp32, xyz [B,N,3], ctr [B,S,3]).  convs[0] contributes only its bias (its weight
    went into u, v, wx).  Returns [B*S*ns, C] rows or, with pool = ns, [B*S, C]."""
    B, S, ns = idx.shape
    if pool not in (0, ns):
        raise ValueError("a gathered stack pools over its own neighbour axis")
    flat = []
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        flat += _layer_args(conv, bn, with_weight=bool(i), m=_MODES["bf16"])
    return _GatheredStack.apply(u.contiguous(), None if v is None else v.contiguous(),
                                wx,
                                None if wx is None else xyz.contiguous(), None if wx is None else ctr.contiguous(),
                                idx.contiguous(), ns, act, pool, len(convs), *flat)


_GATHERED = True


def set_gathered(flag):
    """Enable / disable the gathered first layer (A/B timing and the equivalence tests)."""
    global _GATHERED
    _GATHERED = bool(flag)


def gathered_ok(convs, bns):
    """The gathered first layer needs bf16 rows, 8-aligned layer widths and plain BatchNorm layers."""
    return (_GATHERED and is_bf16() and _stack_fusable(convs, bns)
            and not any(isinstance(b, nn.SyncBatchNorm) for b in bns))


def _stack_desc(layers, widths, ybuf, R, m, outs=None, ext=None):
    """Host descriptor table of pcb_mlp_stack_forward/backward: 18 int64 per layer (pcb_hip.h).  ext: layer 0's
    slot [15] (a ctypes array of 7 int64 the caller keeps alive over the call) or None."""
    vals, yoff = [], 0
    ybase = ybuf.data_ptr()
    esize = ybuf.element_size()
    for l, t in enumerate(layers):
        w, bias, gamma, beta, rm, rv, training = t[:7]
        sliced = (ext is not None and l == 0 and w is not None and w.dim() == 2 and w.stride(1) == 1
                  and w.stride(0) >= w.shape[1])
        if w is not None and not ((w.is_contiguous() or sliced) and w.dtype == torch.float32):
            raise TypeError("fused layers expect contiguous fp32 master weights")
        C = widths[l]
        o = outs[l] if outs is not None else (None, None, None, None)
        vals += [0 if w is None else w.data_ptr(),
                 0 if bias is None else bias.data_ptr(), 0 if gamma is None else gamma.data_ptr(),
                 0 if beta is None else beta.data_ptr(), 0 if rm is None else rm.data_ptr(),
                 0 if rv is None else rv.data_ptr(), C, 0 if w is None else w.numel() // C, int(bool(training)),
                 ybase + esize * yoff,
                 0 if o[0] is None else o[0].data_ptr(), 0 if o[1] is None else o[1].data_ptr(),
                 0 if o[2] is None else o[2].data_ptr(), 0 if o[3] is None else o[3].data_ptr(),
                 t[9].data_ptr() if (len(t) > 9 and t[9] is not None) else 0,
                 ctypes.addressof(ext) if (ext is not None and l == 0) else 0,
                 t[10].data_ptr() if (len(t) > 10 and t[10] is not None) else 0,
                 int(t[11]) if len(t) > 11 else 0]
        yoff += R * C
    return (ctypes.c_longlong * len(vals))(*vals)


def _stack_fusable(convs, bns):
    q = mode().q
    return all(c.out_channels % q == 0 and c.out_channels <= 256 * q for c in convs)


def _fused_stack(convs, bns, x, act, pool, perm, stat_repeat=1):
    m = mode()
    kp = x.shape[1] if perm != 0 else m.pad(convs[0].in_channels)
    xr = x if (x.dtype == m.dtype and x.shape[1] == kp and x.is_contiguous()) else _rows(x, kp, m)
    use_sync, group = _sync_group(bns)
    flat = []
    # num_batches_tracked += 1 (nn.BatchNorm.forward does it per module) rides along in the layer's
    # finalize kernel; the count itself is only read on the host when momentum=None
    for conv, bn in zip(convs, bns):
        flat += _layer_args(conv, bn, m=m, sync=use_sync)
    return _FusedStack.apply(xr, act, pool, perm, stat_repeat, len(convs), m, group if use_sync else False, *flat)


# ---------------------------------------------------------------------------------------------
# public helpers used by the modules
# ---------------------------------------------------------------------------------------------
def conv_bn_act(conv, bn, x, act=ACT_RELU, pool=0, perm=0, stat_repeat=1):
    """act(bn(conv(x))) on rows x [R, K]; with pool = ns also the max over each ns consecutive rows.

    perm = C > 0: x was written by group_rows (C feature columns first, then the 3 centred
    coordinates); perm = -D < 0: by interpolate_concat.  stat_repeat = r > 1: every row stands for r
    identical samples, which only matters for the unbiased running-variance factor.
    Returns [R, Cout] or [R/pool, Cout] in the mode's dtype."""
    if _stack_fusable([conv], [bn]):
        return _fused_stack([conv], [bn], x, act, pool, perm, stat_repeat)
    # widths that are not multiples of a 16-byte chunk (not used by the reference's networks): ATen ops
    if perm != 0 or stat_repeat != 1:
        raise NotImplementedError("column layouts / repeated rows need a channel count the engine takes")
    m = mode()
    y = F.linear(x.to(m.dtype), _weight2d(conv).to(m.dtype), None if conv.bias is None else conv.bias.to(m.dtype))
    y = _act_torch(_bn_rows_fp32(bn, y.float()), act).to(m.dtype)
    if pool:
        y = y.view(-1, pool, y.shape[1]).max(dim=1)[0]
    return y


_gap_rows = {}


def _gap_row_index(n, d, q, device):
    """Row numbers of the n real outputs inside the interpolate+concat column layout (cached)."""
    key = (n, d, q, device)
    if key not in _gap_rows:
        dp = (d + q - 1) // q * q
        _gap_rows[key] = torch.cat([torch.arange(d, device=device), dp + torch.arange(n - d, device=device)])
    return _gap_rows[key]


def ungap_rows(x, perm):
    """Rows in the interpolate_concat column layout (perm = -D: a gap after the first D columns) ->
    the same rows with the real columns only, in reference order."""
    if perm >= 0:
        return x
    m = mode()
    d = -perm
    n = x.shape[1] - (m.pad(d) - d)
    return x.index_select(1, _gap_row_index(n, d, m.q, x.device))


class _LinearBias(torch.autograd.Function):
    """y = x W^T + b on rows (no BatchNorm): operands prepared by one kernel, the product with
    the bias added in the GEMM epilogue.  Backward: input gradient with the same GEMM on W^T, weight
    gradient with the split-row MFMA kernel (fp32, no atomics), bias gradient as a column sum."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_gap, m, res=None, out_dtype=None):
        """out_gap = D > 0: the output keeps the interpolate_concat column layout (first D outputs
        in place, the rest from column pad(D); untouched columns are exactly zero) and is returned
        with all its padded columns; out_gap = 0: plain [R, n] output.
        res (bf16 rows [R, n], n a multiple of 8, out_gap = 0): added to the rounded result in the GEMM's
        epilogue -- the sum of two branches' outputs without an addition pass."""
        w = weight.reshape(weight.shape[0], -1)
        if not (w.is_contiguous() and w.dtype == torch.float32):
            raise TypeError("fused layers expect contiguous fp32 master weights")
        n, k = w.shape
        R, kp = x.shape
        dev = x.device
        npad = m.pad(m.pad(out_gap) + n - out_gap) if out_gap else m.pad(n)
        need_dx = ctx.needs_input_grad[0]
        cache_key = hit = None
        if not torch.is_grad_enabled():  # constant weights: keep the prepared operands (see _eval_lookup)
            cache_key, versions, hit = _eval_lookup([(weight, bias)], weight, ("linear", kp, int(out_gap), m.code))
        if hit is not None:
            wp, wt, bp = hit
        else:
            wp = torch.empty(npad, kp, dtype=m.dtype, device=dev)
            wt = torch.empty(kp, npad, dtype=m.dtype, device=dev) if need_dx else None
            bp = torch.empty(npad, dtype=torch.float32, device=dev)
        y = torch.empty(R, npad, dtype=m.dtype, device=dev)
        with on_device(dev):
            if hit is None:
                _launch("pcb_prep_linear_bias_" + m.sfx, npad * kp, w.data_ptr(), 0 if bias is None else bias.data_ptr(), n, k,
                        npad, kp, int(out_gap), wp.data_ptr(), 0 if wt is None else wt.data_ptr(), bp.data_ptr())
            if res is not None:
                _launch("pcb_gemm_nt_bias_add_bf16", 2 * R * (2 * npad + kp), x.data_ptr(), wp.data_ptr(), bp.data_ptr(),
                        res.data_ptr(), R, npad, kp, y.data_ptr())
            else:
                _launch("pcb_gemm_nt_bias_" + m.sfx, 2 * R * (npad + kp), x.data_ptr(), wp.data_ptr(), bp.data_ptr(), R, npad, kp,
                        y.data_ptr())
        if cache_key is not None and hit is None:
            _eval_store(cache_key, weight, versions, (wp, wt, bp))
        ctx.save_for_backward(x, wt)
        ctx.cfg = (weight.shape, n, k, bias is not None, int(out_gap), npad, m)
        ctx.has_res = res is not None
        out = y if out_gap else y[:, :n]
        # (out_dtype inside the Function: the backward pass then gets the loss's fp32 rows and casts + pads them in one
        # launch, instead of autograd's cast node followed by a pad)
        return out if out_dtype is None or out.dtype == out_dtype else out.to(out_dtype)

    @staticmethod
    def backward(ctx, g):
        x, wt = ctx.saved_tensors
        wshape, n, k, has_bias, out_gap, npad, m = ctx.cfg
        R, kp = x.shape
        dev = x.device
        if g.dtype == torch.float32 and g.dim() == 2 and g.stride(1) == 1 and (g.shape[1] != npad or not g.is_contiguous()):
            # fp32 gradient rows of the real width (the loss's): cast + pad in one pass
            gy = torch.empty(R, npad, dtype=m.dtype, device=dev)
            with on_device(dev):
                _launch("pcb_pad_rows_" + m.sfx, R * npad, g.data_ptr(), g.stride(0), R, g.shape[1], npad, gy.data_ptr())
        else:
            gy = g.to(m.dtype)
            gy = gy.contiguous() if gy.shape[1] == npad else F.pad(gy, (0, npad - gy.shape[1]))
        dx = None
        dw = torch.empty(npad, k, dtype=torch.float32, device=dev)
        lib = _lib.load()
        ws = torch.empty(lib.pcb_gemm_tn_workspace(R, npad, kp), dtype=torch.float32, device=dev)
        fused_bias = has_bias and m.sfx == "bf16"   # the bias gradient out of the weight-gradient pass over dy
        db = torch.empty(npad, dtype=torch.float32, device=dev) if fused_bias else None
        sums = torch.zeros(2, npad, dtype=torch.float32, device=dev) if has_bias and not fused_bias else None
        with on_device(dev):
            if wt is not None:
                dx = torch.empty(R, kp, dtype=m.dtype, device=dev)
                _launch("pcb_gemm_nt_" + m.sfx, 2 * R * (npad + kp), 0, gy.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 0, 0, wt.data_ptr(),
                        R, kp, npad, dx.data_ptr(), 0, 0)
            # dW straight in the real [npad, k] layout (the padding columns of x dropped)
            if fused_bias:
                _launch("pcb_gemm_tn_bias_bf16", 2 * R * (npad + kp), gy.data_ptr(), x.data_ptr(), R, npad, kp, ws.data_ptr(),
                        dw.data_ptr(), k, 0, db.data_ptr())
            else:
                _launch("pcb_gemm_tn_" + m.sfx, 2 * R * (npad + kp), 0, gy.data_ptr(), 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, x.data_ptr(),
                        0, 0, 0, R, npad, kp, ws.data_ptr(), dw.data_ptr(), k, 0)
                if has_bias and _ops.deterministic():
                    nparts = max(1, min(256, R // 64))
                    slabs = torch.empty(nparts, 2, npad, dtype=torch.float32, device=dev)
                    _launch("pcb_colstats_slabs_" + m.sfx, R * npad, gy.data_ptr(), R, npad, slabs.data_ptr(), nparts)
                    _launch("pcb_sum_slabs", 2 * npad, slabs.data_ptr(), nparts, 2 * npad, sums.data_ptr())
                    db = sums[0]
                elif has_bias:
                    _launch("pcb_colstats_" + m.sfx, R * npad, gy.data_ptr(), R, npad, sums.data_ptr())
                    db = sums[0]
        gres = g if ctx.has_res else None   # d(out)/d(res) = identity
        if not out_gap:
            return dx, dw[:n].reshape(wshape), ((db if n == npad else db[:n].clone()) if has_bias else None), None, None, gres, None
        rows = _gap_row_index(n, out_gap, m.q, dev)
        return dx, dw[rows].reshape(wshape), (db[rows] if has_bias else None), None, None, gres, None


def conv_rows(conv, x, out_dtype=None, out_gap=0, add=None):
    """Plain 1x1 conv (with bias) on rows, no BatchNorm; out_gap: see _LinearBias.forward.
    add: rows [R, n] to add to the result; bf16 rows of a width the engine takes go into the GEMM's epilogue."""
    m = mode()
    xr = _rows(x, m.pad(x.shape[1]), m)
    n = conv.out_channels
    if (add is not None and m.sfx == "bf16" and not out_gap and n % m.q == 0 and add.dtype == m.dtype
            and add.shape == (xr.shape[0], n) and add.is_contiguous()):
        return _LinearBias.apply(xr, conv.weight, conv.bias, out_gap, m, add, out_dtype)
    if add is None:
        return _LinearBias.apply(xr, conv.weight, conv.bias, out_gap, m, None, out_dtype)
    y = _LinearBias.apply(xr, conv.weight, conv.bias, out_gap, m) + add
    return y if out_dtype is None else y.to(out_dtype)


def build_interp_csr(idx, S):
    """Inverted index of an interpolation's neighbour table idx [B,N,k] (values in [0,S)): for every
    source row the list of (target, slot) entries that read it -- (offsets [B*S+1] int64, entries
    [B*N*k] int32).  Depends on the coordinates only, so a prefetch can build it ahead."""
    B, N, k = idx.shape
    dev = idx.device
    if _ops.deterministic():
        # reproducible mode: entries of a segment in ascending order (the counting sort below places them in the order
        # its atomic cursors are served, and the consumer adds in entry order)
        order, offsets = _ops.det_index(idx, S)
        return offsets, (order % (N * k)).to(torch.int32)
    count = torch.zeros(2, B * S, dtype=torch.int32, device=dev)  # counts | placement cursors
    entries = torch.empty(B * N * k, dtype=torch.int32, device=dev)
    with on_device(dev):
        _launch("pcb_interp_csr_count", B * N * k, idx.data_ptr(), B, N, S, k, count[0].data_ptr())
        offsets = torch.zeros(B * S + 1, dtype=torch.int64, device=dev)
        offsets[1:] = torch.cumsum(count[0], dim=0)
        _launch("pcb_interp_csr_fill", B * N * k, idx.data_ptr(), B, N, S, k, offsets.data_ptr(),
                count[1].data_ptr(), entries.data_ptr())
    return offsets, entries


class _InterpConcat(torch.autograd.Function):
    """[skip | interpolated] rows of FeaturePropagation (models/pointnet2_utils.py:191-203) written
    once, as the input buffer of the following GEMM: columns [0:D1) = skip features,
    [pad(D1) : pad(D1)+C) = inverse-distance interpolation of feat over the k nearest (from
    three_nn), everything else zero.  Backward: the interpolation's gradient is reduced per target
    row over an inverted index (no atomics)."""

    @staticmethod
    def forward(ctx, skip, feat, d2, idx, offsets, entries, m):
        B, S, C = feat.shape
        N, k = d2.shape[1], d2.shape[2]
        D1 = 0 if skip is None else skip.shape[1]
        dp = m.pad(D1)
        dev = feat.device
        out = torch.empty(B * N, dp + C, dtype=m.dtype, device=dev)
        fused_skip = skip is not None and skip.dtype == m.dtype and skip.dim() == 2 and skip.stride(1) == 1
        if skip is not None and not fused_skip:
            out[:, :D1] = skip
            if dp > D1:
                out[:, D1:dp] = 0
        w = torch.empty(B, N, k, dtype=torch.float32, device=dev)
        with on_device(dev):
            if fused_skip:   # the skip columns and the gap from the same launch (two strided ATen copies otherwise)
                _launch("pcb_interpolate_skip_bf16" if m.code == 0 else "pcb_interpolate_rows_skip_f32", 2 * B * N * C,
                        feat.data_ptr(), d2.data_ptr(), idx.data_ptr(), B, N, S, C, k, out.data_ptr(), dp + C, dp, w.data_ptr(),
                        skip.data_ptr(), skip.stride(0), D1)
            else:
                _launch("pcb_interpolate_bf16" if m.code == 0 else "pcb_interpolate_rows_f32", 2 * B * N * C, feat.data_ptr(),
                        d2.data_ptr(), idx.data_ptr(), B, N, S, C, k, out.data_ptr(), dp + C, dp, w.data_ptr())
        ctx.save_for_backward(idx, w, offsets, entries)
        ctx.shape = (B, N, S, C, k, D1, dp, m)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, w, offsets, entries = ctx.saved_tensors
        B, N, S, C, k, D1, dp, m = ctx.shape
        dev = g.device
        g = g.to(m.dtype).contiguous()
        if offsets is None:  # not built ahead by a prefetch
            offsets, entries = build_interp_csr(idx, S)
        gfeat = torch.empty(B, S, C, dtype=m.dtype, device=dev)
        with on_device(dev):
            _launch("pcb_interpolate_bwd_csr_" + m.sfx, 2 * B * N * C * k, g.data_ptr(), dp + C, dp, w.data_ptr(),
                    offsets.data_ptr(), entries.data_ptr(), B, N, S, C, k, gfeat.data_ptr())
        gskip = g[:, :D1] if (D1 and ctx.needs_input_grad[0]) else None
        return gskip, gfeat, None, None, None, None, None


def interpolate_concat(skip_rows, feat_bsc, d2, idx, csr=None):
    """(rows [B*N, pad(D1)+C], perm) with perm = -D1 describing the column layout for conv_bn_act /
    mlp_rows / conv_rows (0 when there is no gap).  Needs C to be a multiple of the 16-byte chunk.
    csr = build_interp_csr(idx, S) if the caller already has it (the backward pass builds it otherwise)."""
    m = mode()
    feat = feat_bsc.to(m.dtype).contiguous()
    skip = None if skip_rows is None else skip_rows.to(m.dtype)
    offsets, entries = csr if csr is not None else (None, None)
    rows = _InterpConcat.apply(skip, feat, d2.contiguous(), idx.contiguous(), offsets, entries, m)
    d1 = 0 if skip is None else skip.shape[1]
    return rows, (-d1 if d1 % m.q else 0)


class _BNActRows(torch.autograd.Function):
    """act(BatchNorm(y)) on rows [R,C] with no GEMM in front (DGCNN.local_bn): one statistics
    pass, the finalize kernel, one apply pass; backward = one reduce + one apply pass (csrc/rowbn.hip)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, momentum, eps, act, m, nbt):
        R, C = y.shape
        dev = y.device
        stats = torch.zeros(6, C, dtype=torch.float32, device=dev)  # sums(2) | scale | shift | mean | invstd
        out = torch.empty(R, C, dtype=m.dtype, device=dev)
        if training:
            note_parameter_update(weights=False)   # running statistics only: prepared operands of other stacks stay valid
        sums, nparts = stats[0:2], 1
        with on_device(dev):
            if training and _ops.deterministic():
                # reproducible mode: one slab per workgroup, added in slab order by the finalize kernel (no atomics)
                nparts = max(1, min(256, R // 64))
                sums = torch.empty(nparts, 2, C, dtype=torch.float32, device=dev)
                _launch("pcb_colstats_slabs_" + m.sfx, R * C, y.data_ptr(), R, C, sums.data_ptr(), nparts)
            elif training:
                _launch("pcb_colstats_" + m.sfx, R * C, y.data_ptr(), R, C, stats[0:2].data_ptr())
            _launch("pcb_bn_finalize", C, sums.data_ptr(), nparts, R, 0, C,
                    0 if gamma is None else gamma.data_ptr(), 0 if beta is None else beta.data_ptr(), 0,
                    0 if running_mean is None else running_mean.data_ptr(),
                    0 if running_var is None else running_var.data_ptr(),
                    float(momentum), float(eps), int(training), stats[2].data_ptr(), stats[3].data_ptr(),
                    stats[4].data_ptr(), stats[5].data_ptr(), 0 if nbt is None else nbt.data_ptr())
            _launch("pcb_bn_act_" + m.sfx, R * C, y.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), R, C, act,
                    out.data_ptr())
        ctx.save_for_backward(y, stats)
        ctx.cfg = (int(training), act, gamma is not None, m)
        return out

    @staticmethod
    def backward(ctx, g):
        y, stats = ctx.saved_tensors
        training, act, has_affine, m = ctx.cfg
        R, C = y.shape
        dev = y.device
        bsums = torch.zeros(2, C, dtype=torch.float32, device=dev)
        dy = torch.empty(R, C, dtype=m.dtype, device=dev)
        gb = g.to(m.dtype).contiguous()
        with on_device(dev):
            if _ops.deterministic():
                # the reduction in slab form, its totals in slab order, then the apply pass (no atomics)
                nparts = max(2, min(256, R // 64))
                slabs = torch.empty(nparts, 2, C, dtype=torch.float32, device=dev)
                _launch("pcb_bn_act_bwd_reduce_" + m.sfx, R * C, gb.data_ptr(), y.data_ptr(), stats[2].data_ptr(),
                        stats[3].data_ptr(), stats[4].data_ptr(), stats[5].data_ptr(), R, C, act, slabs.data_ptr(), nparts)
                _launch("pcb_sum_slabs", 2 * C, slabs.data_ptr(), nparts, 2 * C, bsums.data_ptr())
                _launch("pcb_bn_act_bwd_apply_" + m.sfx, R * C, gb.data_ptr(), y.data_ptr(), stats[2].data_ptr(),
                        stats[3].data_ptr(), stats[4].data_ptr(), stats[5].data_ptr(), bsums.data_ptr(), R, C, act, training,
                        dy.data_ptr())
            else:
                _launch("pcb_bn_act_bwd_" + m.sfx, R * C, gb.data_ptr(), y.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
                        stats[4].data_ptr(), stats[5].data_ptr(), R, C, act, training, bsums.data_ptr(), dy.data_ptr())
        return (dy, bsums[1] if has_affine else None, bsums[0] if has_affine else None,
                None, None, None, None, None, None, None, None)


class _Gate(torch.autograd.Function):
    """x * sigmoid(a) on rows in one pass each way (pcb_gate_* / pcb_gate_bwd_*)."""

    @staticmethod
    def forward(ctx, x, a, m):
        out = torch.empty_like(x)
        with on_device(x.device):
            _launch("pcb_gate_" + m.sfx, x.numel(), x.data_ptr(), a.data_ptr(), out.data_ptr(), x.numel())
        ctx.save_for_backward(x, a)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, g):
        x, a = ctx.saved_tensors
        m = ctx.m
        g = g.to(m.dtype).contiguous()
        dx, da = torch.empty_like(x), torch.empty_like(a)
        with on_device(x.device):
            _launch("pcb_gate_bwd_" + m.sfx, 3 * x.numel(), g.data_ptr(), x.data_ptr(), a.data_ptr(), dx.data_ptr(),
                    da.data_ptr(), x.numel())
        return dx, da, None


class _DropoutRows(torch.autograd.Function):
    """nn.Dropout on rows with a stateless mask (pcb_dropout_rows_*): forward and backward are the same launch with the
    same seed; the seed is one int64 drawn on the device by torch's generator (under a captured graph that draw is the
    graph-safe one: every replay gets a fresh seed)."""

    @staticmethod
    def forward(ctx, x, p, m):
        seed = torch.randint(0, 1 << 62, (1,), dtype=torch.int64, device=x.device)
        out = torch.empty_like(x)
        with on_device(x.device):
            _launch("pcb_dropout_rows_" + m.sfx, x.numel(), x.data_ptr(), x.numel(), seed.data_ptr(), float(p), out.data_ptr())
        ctx.save_for_backward(seed)
        ctx.cfg = (float(p), m)
        return out

    @staticmethod
    def backward(ctx, g):
        (seed,) = ctx.saved_tensors
        p, m = ctx.cfg
        g = g.to(m.dtype).contiguous()
        dx = torch.empty_like(g)
        with on_device(g.device):
            _launch("pcb_dropout_rows_" + m.sfx, g.numel(), g.data_ptr(), g.numel(), seed.data_ptr(), p, dx.data_ptr())
        return dx, None, None


def dropout_rows(drop, x):
    """drop(x) for an nn.Dropout module and rows x of the mode's type: the library's stateless-mask kernel in training
    mode (one pass forward, one backward, no mask tensor), identity in eval mode; other dtypes / layouts go to the module."""
    m = mode()
    if not drop.training or drop.p == 0.0:
        return x
    if (x.is_cuda and x.dtype == m.dtype and x.is_contiguous() and x.numel() % m.q == 0 and 0.0 < drop.p < 1.0
            and os.environ.get("PCB_DROPOUT_KERNEL", "1") != "0"):    # (A/B knob: tools/ab_env.sh)
        return _DropoutRows.apply(x, drop.p, m)
    return drop(x)


def gate_rows(x, a):
    """x * sigmoid(a) (the channel-attention gate, reference pointnet2_utils.py:279-280)."""
    m = mode()
    if (x.dtype == m.dtype and a.dtype == m.dtype and x.shape == a.shape
            and x.is_contiguous() and a.is_contiguous() and x.numel() % m.q == 0):
        return _Gate.apply(x, a, m)
    return x * torch.sigmoid(a)


class _SceneMax(torch.autograd.Function):
    """max over the N rows of every scene (DGCNN's adaptive_max_pool1d, models/DGCNN.py:160) with own
    kernels (csrc/scenepool.hip): int32 row indices, a dense one-pass backward."""

    @staticmethod
    def forward(ctx, rows, B, N, m):
        C = rows.shape[1]
        dev = rows.device
        out = torch.empty(B, C, dtype=m.dtype, device=dev)
        arg = torch.empty(B, C, dtype=torch.int32, device=dev)
        ws = torch.empty(_lib.load().pcb_scene_max_workspace(B, C), dtype=torch.uint8, device=dev)
        with on_device(dev):
            _launch("pcb_scene_max_" + m.sfx, B * N * C, rows.data_ptr(), B, N, C, out.data_ptr(), arg.data_ptr(), ws.data_ptr())
        ctx.save_for_backward(arg)
        ctx.cfg = (B, N, C, m)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        B, N, C, m = ctx.cfg
        g = g.to(m.dtype).contiguous()
        dz = torch.empty(B * N, C, dtype=m.dtype, device=g.device)
        with on_device(g.device):
            _launch("pcb_scene_max_bwd_" + m.sfx, B * N * C, g.data_ptr(), arg.data_ptr(), B, N, C, dz.data_ptr())
        return dz, None, None, None


def scene_max(rows, B, N):
    """rows [B*N, C] -> [B, C]: the maximum over the N rows of each scene (ties: lowest row)."""
    m = mode()
    if rows.shape[1] % m.q:
        return rows.view(B, N, -1).max(dim=1)[0]
    return _SceneMax.apply(rows.to(m.dtype).contiguous(), B, N, m)


class _SceneConcat(torch.autograd.Function):
    """[a | g broadcast over the scene's rows] with own kernels in both directions (csrc/scenepool.hip)."""

    @staticmethod
    def forward(ctx, a, g, B, N, m):
        C1, C2 = a.shape[1], g.shape[1]
        out = torch.empty(B * N, C1 + C2, dtype=m.dtype, device=a.device)
        with on_device(a.device):
            _launch("pcb_scene_concat_" + m.sfx, B * N * (C1 + C2), a.data_ptr(), g.data_ptr(), B, N, C1, C2, out.data_ptr())
        ctx.cfg = (B, N, C1, C2, m)
        return out

    @staticmethod
    def backward(ctx, d):
        B, N, C1, C2, m = ctx.cfg
        d = d.to(m.dtype).contiguous()
        dg = None
        if ctx.needs_input_grad[1]:
            dg = torch.empty(B, C2, dtype=m.dtype, device=d.device)
            ws = torch.empty(_lib.load().pcb_scene_colsum_workspace(B, N, C2), dtype=torch.uint8, device=d.device)
            with on_device(d.device):
                _launch("pcb_scene_colsum_" + m.sfx, B * N * C2, d.data_ptr(), B, N, C1 + C2, C1, C2, dg.data_ptr(), ws.data_ptr())
        return (d[:, :C1] if ctx.needs_input_grad[0] else None), dg, None, None, None


def scene_concat(a_rows, g, B, N):
    """a_rows [B*N, C1], g [B, C2] -> [B*N, C1+C2]: every row of scene b followed by g[b]
    (DGCNN.py:160-164: the pooled feature expanded over the points and concatenated)."""
    m = mode()
    if a_rows.shape[1] % m.q or g.shape[1] % m.q:
        return torch.cat([a_rows.view(B, N, -1), g.view(B, 1, -1).expand(-1, N, -1).to(a_rows.dtype)], dim=2).view(B * N, -1)
    return _SceneConcat.apply(a_rows.to(m.dtype).contiguous(), g.to(m.dtype).contiguous(), B, N, m)


class _RepeatConcat(torch.autograd.Function):
    """out [rows, sum C_l] = levels concatenated along channels, level l given as [rows / r_l, C_l] rows and
    repeated r_l times (csrc/scenepool.hip: one pass forward, one pass backward with the sums over the repeats)."""

    @staticmethod
    def forward(ctx, reps, m, *levels):
        dev = levels[0].device
        rows = levels[0].shape[0] * reps[0]
        widths = [int(o.shape[1]) for o in levels]
        n = len(levels)
        out = torch.empty(rows, sum(widths), dtype=m.dtype, device=dev)
        src = (ctypes.c_void_p * n)(*[o.data_ptr() for o in levels])
        rep = (ctypes.c_int * n)(*reps)
        wid = (ctypes.c_int * n)(*widths)
        with on_device(dev):
            _launch("pcb_repeat_concat_" + m.sfx, 2 * out.numel() * (16 // m.q), n, src, rep, wid, rows, out.data_ptr())
        ctx.cfg = (reps, widths, m, rows)
        return out

    @staticmethod
    def backward(ctx, g):
        reps, widths, m, rows = ctx.cfg
        dev = g.device
        n = len(widths)
        g = g.to(m.dtype).contiguous()
        grads = [torch.empty(rows // reps[l], widths[l], dtype=m.dtype, device=dev) if ctx.needs_input_grad[2 + l] else None
                 for l in range(n)]
        dst = (ctypes.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in grads])
        rep = (ctypes.c_int * n)(*reps)
        wid = (ctypes.c_int * n)(*widths)
        with on_device(dev):
            _launch("pcb_repeat_concat_bwd_" + m.sfx, g.numel() * (16 // m.q), n, g.data_ptr(), rep, wid, rows, dst)
        return (None, None, *grads)


def repeat_concat(levels, reps):
    """torch.cat([level_l repeated reps[l] times along rows], dim=1) for row tensors [rows / reps[l], C_l] (C_l a
    multiple of the row mode's column quantum): the nearest-neighbour upsampling + concatenation of
    MultiScaleFeatureFusion (models/model.py:150-170) without the per-level copies."""
    m = mode()
    return _RepeatConcat.apply(tuple(int(r) for r in reps), m, *[o.to(m.dtype).contiguous() for o in levels])


def conv_bn_act_levels(conv, bn, levels, reps, act=ACT_RELU, concat=None):
    """act(bn(conv(repeat_concat(levels, reps)))) -- MultiScaleFeatureFusion's upsample + concatenate followed by
    final_fusion's first Conv1d + BatchNorm1d (models/model.py:150-170, :93-99) -- WITHOUT the concatenated rows.

    The conv is linear: its product with [level_a repeated | level_b repeated | full] is
    full W_f^T + repeat(level_a W_a^T) + repeat(level_b W_b^T).  The coarse levels' shares are formed on their own rows
    (point_linear, fp32, reps times fewer rows) and enter the full-resolution GEMM as addends of its accumulators
    (pcb_gemm_nt_stats_add_bf16): the [R, sum C] tensor is neither written nor read, the GEMM and both of its gradient
    GEMMs shrink to the full level's columns, and the gradient of a coarse level is a sum of dy over its repeats
    (pcb_dy_repeat_sums_bf16) instead of a [R, sum C] input gradient summed afterwards.  Same arithmetic as the
    concatenated form up to the order of the fp32 additions (one rounding, after the sum).
    Falls back to that form (`concat()` builds its rows; default repeat_concat) where the fused one does not apply
    (fp32 rows, more than two coarse levels, repeats that are not powers of two >= 4)."""
    m = mode()
    widths = [int(o.shape[1]) for o in levels]
    full = [i for i, r in enumerate(reps) if r == 1]
    coarse = sorted((i for i, r in enumerate(reps) if r != 1), key=lambda i: reps[i])
    fused = (m is _MODES["bf16"] and len(full) == 1 and 1 <= len(coarse) <= 2 and _stack_fusable([conv], [bn])
             and all(w % m.q == 0 for w in widths) and sum(widths) == conv.in_channels and levels[0].is_cuda
             and all(reps[i] >= 4 and (reps[i] & (reps[i] - 1)) == 0 for i in coarse)
             and os.environ.get("PCB_LEVELS_FUSED", "1") != "0")    # (A/B knob: tools/ab_env.sh)
    if fused:
        x = levels[full[0]]
        R = x.shape[0]
        fused = all(levels[i].shape[0] * reps[i] == R for i in coarse)
    if not fused:
        return conv_bn_act(conv, bn, repeat_concat(levels, reps) if concat is None else concat(), act)
    pieces = _SplitColsN.apply(_weight2d(conv), tuple(widths))
    adds = [point_linear(levels[i], pieces[i]) for i in coarse] + [None] * (2 - len(coarse))
    shs = [reps[i].bit_length() - 1 for i in coarse]
    shs += [shs[-1]] * (2 - len(shs))
    use_sync, group = _sync_group([bn])
    flat = _layer_args(conv, bn, m=m, sync=use_sync)
    flat[0] = pieces[full[0]]
    xr = x if (x.dtype == m.dtype and x.is_contiguous()) else x.to(m.dtype).contiguous()
    return _FusedStack.apply(xr, act, 0, 0, 1, 1, m, group if use_sync else False, *flat, adds[0], adds[1], shs[0], shs[1])


def bn_act_rows(bn, x, act=ACT_NONE):
    """BatchNorm (+ activation) on rows without a preceding conv (DGCNN.local_bn)."""
    m = mode()
    if isinstance(bn, nn.SyncBatchNorm) or x.shape[1] % m.q != 0 or x.shape[1] > 256 * m.q:
        return _act_torch(_bn_rows_fp32(bn, x.float()), act).to(m.dtype)
    momentum = bn.momentum if bn.momentum is not None else _bn_bookkeeping(bn)
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    track = bn.track_running_stats and bn.running_mean is not None
    return _BNActRows.apply(x.to(m.dtype).contiguous(), bn.weight, bn.bias,
                            bn.running_mean if (track or not training) else None,
                            bn.running_var if (track or not training) else None,
                            training, momentum, bn.eps, act, m, _counter(bn))


def mlp_rows(convs, bns, x, act=ACT_RELU, pool=0, perm=0):
    """A stack of conv_bn_act layers; the last one pools if pool > 0."""
    n = len(convs)
    if _stack_fusable(convs, bns):
        return _fused_stack(list(convs), list(bns), x, act, pool, perm)
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        x = conv_bn_act(conv, bn, x, act, pool if i == n - 1 else 0, perm if i == 0 else 0)
    return x


def group_rows(xyz, new_xyz, feat, idx):
    """Grouped GEMM input rows for a set-abstraction level: pcb_group_rows_* -> [B*S*ns, Kp] in the
    mode's row type, FEATURES FIRST, then the 3 centred coordinates, zero padded to a 16-byte chunk
    (the reference's order is coordinates first, :56 / :347; the weight columns are permuted to match
    when the stack's operands are prepared).  Returns (rows, perm) with perm = C (see conv_bn_act)."""
    m = mode()
    return _GroupRows.apply(xyz, new_xyz, feat, idx, m), (0 if feat is None else feat.shape[2])


class _GroupRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, idx, m):
        B, N, _ = xyz.shape
        S, ns = idx.shape[1], idx.shape[2]
        C = 0 if feat is None else feat.shape[2]
        kp = m.pad(C + 3)
        if feat is not None:
            feat = feat.to(m.dtype).contiguous()
        out = torch.empty(B * S * ns, kp, dtype=m.dtype, device=xyz.device)
        with on_device(xyz.device):
            _launch("pcb_group_rows_" + m.sfx, B * S * ns * kp, xyz.data_ptr(), new_xyz.data_ptr(),
                    0 if feat is None else feat.data_ptr(), idx.data_ptr(), B, N, S, ns, C, kp, out.data_ptr())
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, S, ns, C, kp, m)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, S, ns, C, kp, m = ctx.shape
        if C == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None, None
        g = g.to(m.dtype).contiguous()
        if _ops.deterministic():
            gf = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
            _ops.segment_sum(g.view(B * S * ns, kp), 0, C, *_ops.det_index(idx, N), gf.view(B * N, C))
            return None, None, gf, None, None
        gf = torch.zeros(B, N, C, dtype=torch.float32, device=g.device)
        with on_device(g.device):
            _launch("pcb_group_rows_%s_bwd" % m.sfx, B * S * ns * C, g.data_ptr(), idx.data_ptr(), B, N, S, ns, C, kp,
                    gf.data_ptr())
        return None, None, gf, None, None
