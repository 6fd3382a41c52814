"""Tensor-level operators over libpcb_hip.so.

Index ops (no gradient): furthest_point_sample, ball_query, ball_query2, three_nn, knn.
Differentiable ops (torch.autograd.Function): gather_rows, group_points, three_interpolate,
edge_features.  The north_star names of the PointNet++ CUDA-extension lineage
(furthest_point_sample, ball_query, group_points, three_nn, three_interpolate) do not exist as
symbols in the reference; its Python compositions in Highway_bridge/models/pointnet2_utils.py and
DGCNN.py define their semantics, and each function below cites the lines it replaces.

Every op needs CUDA (ROCm) tensors and the built library; there is no CPU path.
"""
import os

import torch

from . import _lib


def _stream():
    """Raw HIP stream handle of torch's current stream (the direct binding: a quarter of the cost of
    torch.cuda.current_stream().cuda_stream, and this runs once per launch)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _NoContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


def on_device(dev):
    """`with on_device(t.device):` around launches -- torch.cuda.device(dev) only when `dev` is not
    the current device already (one process per GPU: it always is, and entering / leaving the real
    context manager costs ~5 us, ~100 times per training step)."""
    idx = dev.index
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NO_CONTEXT
    return torch.cuda.device(dev)


# Live timing of the roofline kernel family (bench.py): libpcb_hip.so records HIP events on the
# launch stream around every gemm_nt launch (pcb_timer_* in include/pcb_hip.h), whether the launch
# comes from Python or from the native stack runtime.
def kernel_timer_start():
    """Arm the library's gemm_nt timer (clears earlier samples)."""
    _lib.check(_lib.load().pcb_timer_start(), "pcb_timer_start")


def kernel_timer_enable(flag):
    """Pause / resume sampling (bench.py samples one step in five: an event pair around every
    launch costs microseconds of its own, which must not leak into the throughput figure)."""
    _lib.load().pcb_timer_enable(int(bool(flag)))


def kernel_timer_stop():
    """-> (launches, total milliseconds, algorithmic bytes) of the gemm_nt family since
    kernel_timer_start; synchronises the recorded events."""
    import ctypes
    n, ms, by = ctypes.c_long(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    _lib.check(_lib.load().pcb_timer_stop(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)), "pcb_timer_stop")
    return n.value, ms.value, by.value


def kernel_timer_read(category):
    """After kernel_timer_stop: (launches, milliseconds, bytes) of category 0 (gemm_nt family), 1 (farthest
    point sampling) or -1 (every launch of the library since kernel_timer_start: count and algorithmic
    bytes, not event-timed)."""
    import ctypes
    n, ms, by = ctypes.c_long(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    _lib.check(_lib.load().pcb_timer_read(int(category), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)), "pcb_timer_read")
    return n.value, ms.value, by.value


# A kernel running beside the main stream (the FPS pyramid of the next batch, models/pointnet2_utils.
# prefetch_sampling) takes CUs away from the persistent GEMMs of the backward pass; while its event
# is pending the library is told to size those grids for the rest of the chip.
_background = None  # (event, busy_cus)
_hint_now = 0
_capture_calls = 0
_GRAPH_HINT_CALLS = int(os.environ.get("PCB_GRAPH_HINT_CALLS", "10"))
_capture_limit = _GRAPH_HINT_CALLS


def set_background_work(event, busy_cus, calls=None):
    """calls: inside a capture, how many stack calls from here on keep the hint (default PCB_GRAPH_HINT_CALLS)."""
    global _background, _capture_calls, _capture_limit
    _background = (event, int(busy_cus))
    _capture_calls = 0
    _capture_limit = _GRAPH_HINT_CALLS if calls is None else int(calls)


def apply_concurrency_hint():
    """Call before a batch of backward launches: forwards the current state to the library."""
    global _background, _hint_now
    want = 0
    if _background is not None:
        if torch.cuda.is_current_stream_capturing():
            # a captured step runs beside its own FPS pyramid but cannot ask whether that is over: the pyramid
            # lasts ~1.5 ms of a 5 ms backward pass.  Holding the hint for the whole step costs more than it
            # saves (measured: 12.4 against 10.6 ms/step, round 1); holding it for the first n stack calls after
            # the fork -- the backward passes that do run beside the pyramid -- pays: PN2-MSG, one box,
            # n = 0 / 4 / 8 / 12 / 20: 7.63 / 7.59 / 7.46 / 7.47 / 7.57 ms per step (round 2); round 3 (fused layer backward): n = 0 / 4 / 6 / 8 / 10 /
            # 12 / 14 / 18: 6.95 / 6.94 / 6.81 / 6.68 / 6.65 / 6.67 / 6.68 / 6.76 (PCB_GRAPH_HINT_CALLS, default 10).
            global _capture_calls
            _capture_calls += 1
            want = _background[1] if _capture_calls <= _capture_limit else 0
        elif _background[0].query():
            _background = None          # finished: the GPU is ours again
        else:
            want = _background[1]
    if want != _hint_now:
        _lib.load().pcb_set_concurrency_hint(want)
        _hint_now = want


def _launch(name, units, *args, check=None):
    """Call entry point `name` of libpcb_hip.so on the current stream; raise on a bad status.
    (`units` documents the call's work at the call site; timing lives in the library.)
    check: called after the foreign call returns -- re-raises what a host callback of the call caught
    (an exception must not unwind through the C frames)."""
    fn = _entry.get(name)
    if fn is None or _lib._lib is None:  # first use, or the library handle was dropped: (re)load, fail loudly
        fn = _entry[name] = getattr(_lib.load(), name)
    status = fn(*args, torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    if check is not None:
        check()
    if status:
        _lib.check(status, name)


_entry = {}  # bound entry points of libpcb_hip.so


def _need_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "pointcloud_bridge_amd operators run on the GPU only (HIP kernels, no CPU fallback); "
                f"got a tensor on {t.device}")
    dev = tensors[0].device
    for t in tensors:
        if t is not None and t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _i64c(t, name):
    if t.dtype != torch.int64:
        raise TypeError(f"{name} must be int64 (torch.long), got {t.dtype}")
    return t.contiguous()


def _xyz(t, name):
    if t.dim() != 3 or t.shape[-1] != 3:
        raise ValueError(f"{name} must be [B,N,3], got {tuple(t.shape)}")
    return _f32c(t, name)


# --------------------------------------------------------------------------------------------
# index ops
# --------------------------------------------------------------------------------------------
def square_distance(src, dst):
    """Materialised pairwise squared distances (pointnet2_utils.py:7-14) -> [B,N,M] fp32."""
    _need_cuda(src, dst)
    src, dst = _xyz(src, "src"), _xyz(dst, "dst")
    B, N, _ = src.shape
    M = dst.shape[1]
    out = torch.empty(B, N, M, dtype=torch.float32, device=src.device)
    with on_device(src.device):
        _launch("pcb_square_distance", B * N * M, src.data_ptr(), dst.data_ptr(), B, N, M, out.data_ptr())
    return out


def furthest_point_sample(xyz, npoint, start_idx):
    """FPS, replaces pointnet2_utils.py:63-80.  start_idx [B] int64 = the reference's randint draw."""
    _need_cuda(xyz, start_idx)
    xyz = _xyz(xyz, "xyz")
    start_idx = _i64c(start_idx, "start_idx")
    B, N, _ = xyz.shape
    if start_idx.shape != (B,):
        raise ValueError(f"start_idx must be [B]={B}, got {tuple(start_idx.shape)}")
    out = torch.empty(B, int(npoint), dtype=torch.int64, device=xyz.device)
    with on_device(xyz.device):
        _launch("pcb_fps", B * N * int(npoint), xyz.data_ptr(), B, N, int(npoint), start_idx.data_ptr(), out.data_ptr())
    return out


def furthest_point_sample_into(xyz, start_idx, out_idx):
    """pcb_fps writing into a caller-owned [B,S] int64 tensor (static buffers of captured steps)."""
    B, N, _ = xyz.shape
    with on_device(xyz.device):
        _launch("pcb_fps", B * N * out_idx.shape[1], xyz.data_ptr(), B, N, out_idx.shape[1], start_idx.data_ptr(),
                out_idx.data_ptr())
    return out_idx


def gather_rows_into(points, idx, out):
    """pcb_gather_rows (no autograd) writing into a caller-owned [B,M,C] fp32 tensor."""
    B, N, C = points.shape
    M = idx.shape[1]
    with on_device(points.device):
        _launch("pcb_gather_rows", B * M * C, points.data_ptr(), idx.data_ptr(), B, N, C, M, out.data_ptr())
    return out


def _r2(radius):
    # python double square, one rounding to fp32: what ATen does with the scalar in
    # `sqrdists > radius ** 2` (pointnet2_utils.py:105)
    return float(torch.tensor(float(radius) ** 2, dtype=torch.float32).item())


def ball_query(radius, nsample, xyz, new_xyz):
    """Ball query, replaces pointnet2_utils.py:97-112 -> [B,S,nsample] int64."""
    _need_cuda(xyz, new_xyz)
    xyz, new_xyz = _xyz(xyz, "xyz"), _xyz(new_xyz, "new_xyz")
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    nsample = int(nsample)
    if nsample > N:
        # the reference fails here too (mask/tensor shape mismatch at pointnet2_utils.py:110)
        raise IndexError(f"nsample ({nsample}) exceeds the number of points ({N})")
    out = torch.empty(B, S, nsample, dtype=torch.int64, device=xyz.device)
    with on_device(xyz.device):
        _launch("pcb_ball_query", B * S * N, xyz.data_ptr(), new_xyz.data_ptr(), B, N, S, _r2(radius), nsample, out.data_ptr())
    return out


def ball_query2(radii, nsamples, xyz, new_xyz):
    """Two ball queries over the same centroids in one sweep (pointnet2_utils.py:340-341 twice)."""
    _need_cuda(xyz, new_xyz)
    xyz, new_xyz = _xyz(xyz, "xyz"), _xyz(new_xyz, "new_xyz")
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    (ra, rb), (na, nb) = radii, (int(nsamples[0]), int(nsamples[1]))
    if max(na, nb) > N:
        raise IndexError(f"nsample ({max(na, nb)}) exceeds the number of points ({N})")
    oa = torch.empty(B, S, na, dtype=torch.int64, device=xyz.device)
    ob = torch.empty(B, S, nb, dtype=torch.int64, device=xyz.device)
    with on_device(xyz.device):
        _launch("pcb_ball_query2", B * S * N, xyz.data_ptr(), new_xyz.data_ptr(), B, N, S, _r2(ra), na, oa.data_ptr(), _r2(rb), nb, ob.data_ptr())
    return oa, ob


def three_nn(xyz1, xyz2, k=3):
    """k nearest of xyz2 for each point of xyz1 (pointnet2_utils.py:185-188, :253-256).

    Returns (d2 [B,N,k] fp32 ascending, idx [B,N,k] int64)."""
    _need_cuda(xyz1, xyz2)
    xyz1, xyz2 = _xyz(xyz1, "xyz1"), _xyz(xyz2, "xyz2")
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    if not 1 <= k <= 4 or S < k:
        raise ValueError(f"three_nn needs 1 <= k <= 4 and S >= k (k={k}, S={S})")
    d2 = torch.empty(B, N, k, dtype=torch.float32, device=xyz1.device)
    idx = torch.empty(B, N, k, dtype=torch.int64, device=xyz1.device)
    with on_device(xyz1.device):
        _launch("pcb_three_nn", B * N * S, xyz1.data_ptr(), xyz2.data_ptr(), B, N, S, k, d2.data_ptr(), idx.data_ptr())
    return d2, idx


def attention(qkv, num_heads, scale=None):
    """softmax(q k^T * scale) v over all points of a scene, for every head: qkv [B, N, 3*C] -- the output of the
    reference's qkv projection (models/PointTransformerV3.py:96: [.., 3, H, C/H]) -- -> [B, N, C] bf16, the layout
    after the reference's transpose + reshape (:113).  Forward only (cfg5 is an inference configuration);
    head_dim in {64, 128, 192, 256}; scale defaults to head_dim ** -0.5 (:79)."""
    _need_cuda(qkv)
    if qkv.dim() != 3 or qkv.shape[2] % (3 * num_heads):
        raise ValueError(f"qkv must be [B, N, 3*C] with C divisible by num_heads, got {tuple(qkv.shape)}")
    if qkv.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("pcb_attention_fwd has no backward: call it under torch.no_grad()")
    B, N, C3 = qkv.shape
    C = C3 // 3
    D = C // num_heads
    if D not in (64, 128, 192, 256):
        raise ValueError(f"head_dim {D} not supported (64, 128, 192, 256)")
    x = qkv.detach().to(torch.bfloat16).contiguous()
    out = torch.empty(B, N, C, dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        _launch("pcb_attention_fwd_bf16", 4 * B * num_heads * N * N * D, x.data_ptr(), B, N, num_heads, D,
                float(D ** -0.5 if scale is None else scale), out.data_ptr())
    return out


def add_layernorm(x, h, pos, norm, want_sum=False):
    """Token rows x [.., C] bf16: x' = x + h (h may be None), LayerNorm(x') with `norm`'s parameters (+ pos, may be None)
    in one pass (csrc/tokens.hip).  Returns (x' or None, normalised rows); x' only when want_sum and h is given.
    Inference helper of models/PointTransformerV3.py (:119-148): no autograd."""
    _need_cuda(x)
    C = x.shape[-1]
    xr = x.detach().reshape(-1, C)
    R = xr.shape[0]
    for t in (xr, h, pos):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or t.numel() != R * C):
            raise ValueError("add_layernorm: contiguous bf16 rows of one shape")
    g, b = norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous()
    out = torch.empty_like(xr)
    xout = torch.empty_like(xr) if (want_sum and h is not None) else None
    with on_device(x.device):
        _launch("pcb_add_layernorm_bf16", R * C, xr.data_ptr(), 0 if h is None else h.data_ptr(), 0 if pos is None else pos.data_ptr(),
                g.data_ptr(), b.data_ptr(), float(norm.eps), R, C, 0 if xout is None else xout.data_ptr(), out.data_ptr())
    return (None if xout is None else xout.view(x.shape)), out.view(x.shape)


def geglu(y):
    """y [.., 2H] bf16 -> y[.., :H] * gelu(y[.., H:]) in one pass (csrc/tokens.hip; models/PointTransformerV3.py:8-21)."""
    _need_cuda(y)
    H = y.shape[-1] // 2
    yr = y.detach().reshape(-1, 2 * H)
    if yr.dtype != torch.bfloat16 or not yr.is_contiguous() or H % 8:
        raise ValueError("geglu: contiguous bf16 rows, half width a multiple of 8")
    out = torch.empty(yr.shape[0], H, dtype=torch.bfloat16, device=y.device)
    with on_device(y.device):
        _launch("pcb_geglu_bf16", yr.numel(), yr.data_ptr(), yr.shape[0], H, out.data_ptr())
    return out.view(*y.shape[:-1], H)


def knn(x_bnd, k):
    """kNN graph on x [B,N,D] (DGCNN.py:49-70 after its transpose at :60) -> [B,N,k] int64."""
    _need_cuda(x_bnd)
    if x_bnd.dim() != 3:
        raise ValueError(f"x must be [B,N,D], got {tuple(x_bnd.shape)}")
    x = _f32c(x_bnd, "x")
    B, N, D = x.shape
    k = int(k)
    if k > N:
        raise RuntimeError(f"k ({k}) exceeds the number of points ({N})")  # torch.topk raises too
    if not 1 <= k <= 32 or D > 128:
        raise ValueError(f"knn supports 1 <= k <= 32 and D <= 128 (k={k}, D={D})")
    out = torch.empty(B, N, k, dtype=torch.int64, device=x.device)
    norms = torch.empty(B, N, dtype=torch.float32, device=x.device)  # |x|^2 scratch of the kernel
    with on_device(x.device):
        if D == 3 and _GRID_KNN:
            # coordinates: the grid search of csrc/knngrid.hip (same output as the all-pairs kernel)
            ws = torch.empty(_lib.load().pcb_knn_xyz_workspace(B, N), dtype=torch.uint8, device=x.device)
            _launch("pcb_knn_xyz", B * N * k, x.data_ptr(), B, N, k, norms.data_ptr(), ws.data_ptr(), out.data_ptr())
        else:
            nbytes = _lib.load().pcb_knn_screen_workspace(B, N, D, k) if _SCREEN_KNN else 0
            if nbytes:
                # feature-space graphs: screening pass on the bf16 matrix core, exact recheck (same output)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                _launch("pcb_knn_screened", B * N * N, x.data_ptr(), B, N, D, k, norms.data_ptr(), ws.data_ptr(),
                        out.data_ptr())
                if _KNN_STATS is not None:
                    _KNN_STATS.append((B, N, D, k, ws[4 * B:8 * B].view(torch.int32).clone()))
            else:
                _launch("pcb_knn", B * N * N, x.data_ptr(), B, N, D, k, norms.data_ptr(), out.data_ptr())
    return out


_SCREEN_KNN = os.environ.get("PCB_SCREEN_KNN", "1") != "0"  # 0: the exact all-pairs kernel alone (A/B timing)
_KNN_STATS = None  # a list: every screened call appends (B, N, D, k, recomputed queries per scene [B] int32)


def set_screen_knn(flag):
    global _SCREEN_KNN
    old, _SCREEN_KNN = _SCREEN_KNN, bool(flag)
    return old


def collect_knn_stats(flag):
    """Start (True) or stop (False) recording how many queries each screened kNN call handed to the exact kernel;
    returns the list recorded so far."""
    global _KNN_STATS
    old, _KNN_STATS = _KNN_STATS, ([] if flag else None)
    return old


_GRID_KNN = os.environ.get("PCB_GRID_KNN", "1") != "0"  # 0: all-pairs kernel for coordinates too (A/B timing)


def set_grid_knn(flag):
    global _GRID_KNN
    _GRID_KNN = bool(flag)


def structure_features(xyz, idx, with_offsets=True):
    """Neighbourhood descriptor of BridgeStructureEncoding (attention_modules.py:595-603, :620-687):
    xyz [B,N,3], idx [B,N,k] int64 -> (feat [B,N,13], rel [B,N,k,3] or None), fp32, no gradient
    (both are functions of the coordinates only)."""
    _need_cuda(xyz, idx)
    x = _xyz(xyz, "xyz")
    nb = _i64c(idx, "idx")
    B, N, _ = x.shape
    if nb.dim() != 3 or nb.shape[0] != B or nb.shape[1] != N:
        raise ValueError(f"idx must be [B,N,k] for xyz {tuple(x.shape)}, got {tuple(nb.shape)}")
    k = nb.shape[2]
    if not 2 <= k <= 32:
        raise ValueError(f"structure_features supports 2 <= k <= 32 (k={k})")
    feat = torch.empty(B, N, 13, dtype=torch.float32, device=x.device)
    rel = torch.empty(B, N, k, 3, dtype=torch.float32, device=x.device) if with_offsets else None
    with on_device(x.device):
        _launch("pcb_structure_features", B * N * k, x.data_ptr(), nb.data_ptr(), B, N, k,
                feat.data_ptr(), 0 if rel is None else rel.data_ptr())
    return feat, rel


# --------------------------------------------------------------------------------------------
# reproducible mode
# --------------------------------------------------------------------------------------------
# The gradients of the gathers are scatter-adds (the reference's index_put_(accumulate=True) behind index_points,
# the grouping and get_graph_feature).  By default they run as fp32 atomics at memory: the fastest form on this chip,
# but the order of the additions varies from run to run.  In the reproducible mode every backward pass of this package
# sums in a fixed order instead (csrc/segsum.hip over an inverted index, slab sums in the BatchNorm kernels) and two
# runs of one training command give bit-identical gradients.  It is on when set_deterministic(True) was called, when
# the environment says PCB_DETERMINISTIC=1, or when torch.use_deterministic_algorithms(True) is in force.
_deterministic = os.environ.get("PCB_DETERMINISTIC", "0") != "0"


def set_deterministic(flag):
    """Reproducible backward passes on / off for this process; returns the previous setting."""
    global _deterministic
    old, _deterministic = _deterministic, bool(flag)
    return old


def deterministic():
    return _deterministic or torch.are_deterministic_algorithms_enabled()


def det_index(idx, n_targets):
    """Inverted index of a gather whose source rows read target rows idx [B, ...] (values clamped to [0, n_targets),
    per scene): (order int32 [E], offsets int64 [B * n_targets + 1]) -- for global target row t the global source rows
    order[offsets[t] : offsets[t + 1]], ascending (torch.sort with stable=True: rocPRIM's radix sort, deterministic)."""
    B = idx.shape[0]
    flat = idx.reshape(B, -1).clamp(0, n_targets - 1)
    tgt = (flat + torch.arange(B, device=idx.device).view(B, 1) * n_targets).reshape(-1)
    key, order = torch.sort(tgt, stable=True)
    offsets = torch.searchsorted(key, torch.arange(B * n_targets + 1, device=idx.device))
    return order.to(torch.int32), offsets.contiguous()


def segment_sum(rows, col0, C, order, offsets, out, accumulate=False):
    """out[t, :C] (fp32 rows) = (out[t] if accumulate else 0) + sum over the segment of rows[order[e], col0 : col0 + C]
    in index order; rows: 2-D fp32 or bf16 with contiguous rows."""
    sfx = {torch.float32: "f32", torch.bfloat16: "bf16"}[rows.dtype]
    if rows.dim() != 2 or rows.stride(1) != 1 or out.dim() != 2 or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("segment_sum: 2-D rows and fp32 2-D out with contiguous rows")
    with on_device(rows.device):
        _launch("pcb_segment_sum_" + sfx, order.numel() * C, rows.data_ptr(), rows.stride(0), int(col0), int(C), order.data_ptr(),
                offsets.data_ptr(), out.shape[0], out.data_ptr(), out.stride(0), int(bool(accumulate)))
    return out


def sum_slabs(slabs):
    """slabs [n, ...] fp32 -> their sum over the first axis, added in slab order by one small kernel (fp64 accumulation).
    torch.sum(dim=0) is the same number up to rounding, but ATen's reductions go multi-stage (staging buffer +
    semaphore) at sizes it chooses, and those do not survive hipGraph replay on this stack (tools/graph_reduce_repro.py):
    every partial-sum total of this package that may run inside a captured step goes through here."""
    slabs = slabs.contiguous()
    n = slabs.shape[0]
    out = torch.empty(slabs.shape[1:], dtype=torch.float32, device=slabs.device)
    with on_device(slabs.device):
        _launch("pcb_sum_slabs", slabs.numel(), slabs.data_ptr(), n, out.numel(), out.data_ptr())
    return out


# --------------------------------------------------------------------------------------------
# differentiable ops
# --------------------------------------------------------------------------------------------
class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx_flat):
        B, N, C = points.shape
        M = idx_flat.shape[1]
        out = torch.empty(B, M, C, dtype=torch.float32, device=points.device)
        with on_device(points.device):
            _launch("pcb_gather_rows", B * M * C, points.data_ptr(), idx_flat.data_ptr(), B, N, C, M, out.data_ptr())
        ctx.save_for_backward(idx_flat)
        ctx.shape = (B, N, C, M)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx_flat,) = ctx.saved_tensors
        B, N, C, M = ctx.shape
        g = g.contiguous()
        if deterministic():
            gp = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
            segment_sum(g.view(B * M, C), 0, C, *det_index(idx_flat, N), gp.view(B * N, C))
            return gp, None
        gp = torch.zeros(B, N, C, dtype=torch.float32, device=g.device)
        with on_device(g.device):
            _launch("pcb_gather_rows_bwd", B * M * C, g.data_ptr(), idx_flat.data_ptr(), B, N, C, M, gp.data_ptr())
        return gp, None


def gather_rows(points, idx):
    """index_points (pointnet2_utils.py:17-39): points [B,N,C] fp32, idx [B,...] int64 -> [B,...,C].

    Indices are clamped to [0,N-1] like the reference (:34-36)."""
    _need_cuda(points, idx)
    if points.dim() != 3:
        raise ValueError(f"points must be [B,N,C], got {tuple(points.shape)}")
    points = _f32c(points, "points")
    idx = _i64c(idx, "idx")
    B = points.shape[0]
    out = _GatherRows.apply(points, idx.reshape(B, -1))
    return out.view(*idx.shape, points.shape[2])


class _GroupPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, idx):
        B, N, _ = xyz.shape
        S, ns = idx.shape[1], idx.shape[2]
        C = 0 if feat is None else feat.shape[2]
        out = torch.empty(B, S, ns, 3 + C, dtype=torch.float32, device=xyz.device)
        with on_device(xyz.device):
            _launch("pcb_group_points", B * S * ns * (3 + C),  xyz.data_ptr(), new_xyz.data_ptr(), 0 if feat is None else feat.data_ptr(), idx.data_ptr(), B, N, S, ns, C, out.data_ptr())
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, S, ns, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, S, ns, C = ctx.shape
        if C == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None
        g = g.contiguous()
        if deterministic():
            gf = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
            segment_sum(g.view(B * S * ns, 3 + C), 3, C, *det_index(idx, N), gf.view(B * N, C))
            return None, None, gf, None
        gf = torch.zeros(B, N, C, dtype=torch.float32, device=g.device)
        with on_device(g.device):
            _launch("pcb_group_points_bwd", B * S * ns * C, g.data_ptr(), idx.data_ptr(), B, N, S, ns, C, gf.data_ptr())
        return None, None, gf, None


def group_points(xyz, new_xyz, feat, idx):
    """cat(xyz[idx] - new_xyz, feat[idx]) -> [B,S,ns,3+C] (pointnet2_utils.py:51-58, :342-349).

    feat is [B,N,C] fp32 or None.  Gradient flows to feat only: xyz is input data."""
    _need_cuda(xyz, new_xyz, feat, idx)
    xyz, new_xyz = _xyz(xyz, "xyz"), _xyz(new_xyz, "new_xyz")
    idx = _i64c(idx, "idx")
    if feat is not None:
        feat = _f32c(feat, "feat")
        if feat.shape[:2] != xyz.shape[:2]:
            raise ValueError(f"feat {tuple(feat.shape)} does not match xyz {tuple(xyz.shape)}")
    if idx.dim() != 3 or idx.shape[:2] != new_xyz.shape[:2]:
        raise ValueError(f"idx must be [B,S,ns] matching new_xyz, got {tuple(idx.shape)}")
    return _GroupPoints.apply(xyz, new_xyz, feat, idx)


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, d2, idx):
        B, S, C = feat.shape
        N, k = d2.shape[1], d2.shape[2]
        out = torch.empty(B, N, C, dtype=torch.float32, device=feat.device)
        w = torch.empty(B, N, k, dtype=torch.float32, device=feat.device)
        with on_device(feat.device):
            _launch("pcb_interpolate", B * N * C, feat.data_ptr(), d2.data_ptr(), idx.data_ptr(), B, N, S, C, k, out.data_ptr(), w.data_ptr())
        ctx.save_for_backward(w, idx)
        ctx.shape = (B, N, S, C, k)
        return out

    @staticmethod
    def backward(ctx, g):
        w, idx = ctx.saved_tensors
        B, N, S, C, k = ctx.shape
        g = g.contiguous()
        if deterministic() and C % 4 == 0:
            # weighted segment sum over the inverted index, entries ascending (the CSR kernel of the row engine)
            order, offsets = det_index(idx, S)
            entries = (order % (N * k)).to(torch.int32)
            gf = torch.empty(B, S, C, dtype=torch.float32, device=g.device)
            with on_device(g.device):
                _launch("pcb_interpolate_bwd_csr_f32", B * N * C * k, g.data_ptr(), C, 0, w.data_ptr(), offsets.data_ptr(),
                        entries.data_ptr(), B, N, S, C, k, gf.data_ptr())
            return gf, None, None
        gf = torch.zeros(B, S, C, dtype=torch.float32, device=g.device)
        with on_device(g.device):
            _launch("pcb_interpolate_bwd", B * N * C, g.data_ptr(), w.data_ptr(), idx.data_ptr(), B, N, S, C, k, gf.data_ptr())
        return gf, None, None


def three_interpolate(feat_bsc, d2, idx):
    """Inverse-distance weighted sum of the k nearest rows (pointnet2_utils.py:191-196, :259-267).

    feat [B,S,C], d2/idx [B,N,k] from three_nn -> [B,N,C].  Gradient flows to feat only."""
    _need_cuda(feat_bsc, d2, idx)
    feat = _f32c(feat_bsc, "feat")
    d2 = _f32c(d2, "d2")
    idx = _i64c(idx, "idx")
    if d2.shape != idx.shape or d2.dim() != 3 or d2.shape[0] != feat.shape[0]:
        raise ValueError(f"d2 {tuple(d2.shape)} / idx {tuple(idx.shape)} / feat {tuple(feat.shape)} mismatch")
    return _ThreeInterpolate.apply(feat, d2, idx)


class _EdgeFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        B, N, D = x.shape
        k = idx.shape[2]
        out = torch.empty(B, N, k, 2 * D, dtype=torch.float32, device=x.device)
        with on_device(x.device):
            _launch("pcb_edge_features", B * N * k * 2 * D, x.data_ptr(), idx.data_ptr(), B, N, D, k, out.data_ptr())
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, D, k)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, D, k = ctx.shape
        g = g.contiguous()
        if deterministic():
            # centre term: sum over the point's own k rows (a dense reduction); neighbour terms: segment sum
            g4 = g.view(B, N, k, 2 * D)
            gx = (g4[..., D:] - g4[..., :D]).sum(dim=2).contiguous()
            segment_sum(g.view(B * N * k, 2 * D), 0, D, *det_index(idx, N), gx.view(B * N, D), accumulate=True)
            return gx, None
        gx = torch.zeros(B, N, D, dtype=torch.float32, device=g.device)
        with on_device(g.device):
            _launch("pcb_edge_features_bwd", B * N * k * 2 * D, g.data_ptr(), idx.data_ptr(), B, N, D, k, gx.data_ptr())
        return gx, None


def edge_features(x_bnd, idx):
    """EdgeConv input cat(x_j - x_i, x_i) -> [B,N,k,2D] (DGCNN.py:90-107, channels-last)."""
    _need_cuda(x_bnd, idx)
    x = _f32c(x_bnd, "x")
    idx = _i64c(idx, "idx")
    if x.dim() != 3 or idx.dim() != 3 or idx.shape[:2] != x.shape[:2]:
        raise ValueError(f"x {tuple(x.shape)} / idx {tuple(idx.shape)} mismatch")
    return _EdgeFeatures.apply(x, idx)
