"""numpy front end of the CPU oracle (oracle/pcb_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

Parity status: pinned against outputs of the reference generated in the build
container (tests/golden/*.npz, generator tests/golden/make_golden.py).

Each function names the reference code it restates
(/root/reference/Highway_bridge/models/pointnet2_utils.py, .../DGCNN.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpcb_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_int = ctypes.c_int


def build(force=False):
    """Compile libpcb_oracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pcb_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpcb_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_square_distance.argtypes = [_f32p, _f32p, _int, _int, _int, _f32p]
        L.orc_fps.argtypes = [_f32p, _int, _int, _int, _i64p, _i64p]
        L.orc_ball_query.argtypes = [_f32p, _f32p, _int, _int, _int, ctypes.c_float, _int, _i64p]
        L.orc_three_nn.argtypes = [_f32p, _f32p, _int, _int, _int, _int, _f32p, _i64p]
        L.orc_knn.argtypes = [_f32p, _int, _int, _int, _int, _i64p, ctypes.c_void_p]
        L.orc_gather_rows.argtypes = [_f32p, _i64p, _int, _int, _int, _int, _f32p]
        L.orc_interpolate.argtypes = [_f32p, _f32p, _i64p, _int, _int, _int, _int, _int, _f32p, ctypes.c_void_p]
        for f in (L.orc_square_distance, L.orc_fps, L.orc_ball_query, L.orc_three_nn, L.orc_knn,
                  L.orc_gather_rows, L.orc_interpolate):
            f.restype = None
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def square_distance(src, dst):
    """pointnet2_utils.py:7-14 -> [B,N,M] fp32."""
    src, dst = _f32(src), _f32(dst)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = np.empty((B, N, M), np.float32)
    lib().orc_square_distance(src, dst, B, N, M, out)
    return out


def farthest_point_sample(xyz, npoint, start):
    """pointnet2_utils.py:63-80; `start` [B] int64 is the reference's torch.randint draw (:69)."""
    xyz, start = _f32(xyz), _i64(start)
    B, N, _ = xyz.shape
    out = np.empty((B, npoint), np.int64)
    lib().orc_fps(xyz, B, N, npoint, start, out)
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    """pointnet2_utils.py:97-112 -> [B,S,nsample] int64."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, nsample), np.int64)
    r2 = np.float32(radius ** 2)  # python double square, then one rounding to fp32 (ATen scalar cast)
    lib().orc_ball_query(xyz, new_xyz, B, N, S, float(r2), nsample, out)
    return out


def three_nn(xyz1, xyz2, k=3):
    """pointnet2_utils.py:185-188 (k=3) / :253-256 (k=4) -> (d2 [B,N,k] fp32, idx [B,N,k] int64)."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    assert 1 <= k <= 8
    d = np.empty((B, N, k), np.float32)
    i = np.empty((B, N, k), np.int64)
    lib().orc_three_nn(xyz1, xyz2, B, N, S, k, d, i)
    return d, i


def knn(x_bnd, k, return_dist=False):
    """DGCNN.py:49-70 on x already laid out [B,N,D] -> idx [B,N,k] int64 (best first)."""
    x = _f32(x_bnd)
    B, N, D = x.shape
    idx = np.empty((B, N, k), np.int64)
    if return_dist:
        d = np.empty((B, N, k), np.float32)
        lib().orc_knn(x, B, N, D, k, idx, d.ctypes.data_as(ctypes.c_void_p))
        return idx, d
    lib().orc_knn(x, B, N, D, k, idx, None)
    return idx


def index_points(points, idx):
    """pointnet2_utils.py:17-39 (with the clamp of :34-36) for fp32 points [B,N,C]."""
    points, idx = _f32(points), _i64(idx)
    B, N, C = points.shape
    flat = idx.reshape(B, -1)
    out = np.empty((B, flat.shape[1], C), np.float32)
    lib().orc_gather_rows(points, flat, B, N, C, flat.shape[1], out)
    return out.reshape(*idx.shape, C)


def three_interpolate(feat_bsc, d2, idx, return_weight=False):
    """pointnet2_utils.py:191-196: inverse-distance weights and the weighted gather-sum."""
    feat, d2, idx = _f32(feat_bsc), _f32(d2), _i64(idx)
    B, S, C = feat.shape
    N, k = d2.shape[1], d2.shape[2]
    out = np.empty((B, N, C), np.float32)
    if return_weight:
        w = np.empty((B, N, k), np.float32)
        lib().orc_interpolate(feat, d2, idx, B, N, S, C, k, out, w.ctypes.data_as(ctypes.c_void_p))
        return out, w
    lib().orc_interpolate(feat, d2, idx, B, N, S, C, k, out, None)
    return out


def edge_features(x_bnd, idx):
    """DGCNN.py:90-107: [x_j - x_i, x_i] -> [B, 2D, N, k] fp32."""
    x, idx = _f32(x_bnd), _i64(idx)
    nb = index_points(x, idx)  # [B,N,k,D]
    ctr = np.broadcast_to(x[:, :, None, :], nb.shape)
    f = np.concatenate([nb - ctr, ctr], axis=3)
    return np.ascontiguousarray(f.transpose(0, 3, 1, 2))


def cdist_knn(xyz, k):
    """attention_modules.py:584-586: Euclidean distance matrix, k smallest per row -> [B,N,k] int64.
    numpy restatement of torch.cdist's matmul route (|a|^2 + |b|^2 - 2ab, clamped at 0, sqrt);
    candidates tied at the k-th place may differ from ATen's topk -- compare as sets."""
    x = _f32(xyz).astype(np.float64)
    n2 = (x * x).sum(-1)
    d2 = n2[:, :, None] + n2[:, None, :] - 2.0 * np.einsum("bnc,bmc->bnm", x, x)
    d = np.sqrt(np.maximum(d2, 0.0)).astype(np.float32)
    return np.ascontiguousarray(np.argsort(d, axis=-1, kind="stable")[..., :k]).astype(np.int64)


def structure_features(xyz, idx):
    """BridgeStructureEncoding.get_structure_features, attention_modules.py:620-687, on the
    neighbourhoods idx [B,N,k] of xyz [B,N,3] -> (feat [B,N,13], rel [B,N,k,3]) fp32."""
    xyz, idx = _f32(xyz), _i64(idx)
    B, N, k = idx.shape
    rel = (index_points(xyz, idx) - xyz[:, :, None, :]).astype(np.float32)      # :595-600
    second = (np.einsum("bnka,bnkc->bnac", rel, rel) / np.float32(k - 1)).astype(np.float32)
    ev = np.linalg.eigvalsh(second).astype(np.float32)                           # ascending, :634
    den = ev[..., 0] + np.float32(1e-8)
    f = np.empty((B, N, 13), np.float32)
    f[..., 0] = (ev[..., 0] - ev[..., 1]) / den
    f[..., 1] = (ev[..., 1] - ev[..., 2]) / den
    f[..., 2] = ev[..., 2] / den
    mean = rel.mean(axis=2, keepdims=True)
    spread = np.sqrt(((rel - mean) ** 2).sum(-1))                                # :646-647
    f[..., 3] = spread.max(-1)
    f[..., 4] = spread.mean(-1)
    f[..., 5] = spread.std(-1, ddof=1)
    unit = rel / (np.sqrt((rel ** 2).sum(-1, keepdims=True)) + np.float32(1e-8))  # :657
    f[..., 6] = np.einsum("bnja,bnla->bnjl", unit, unit).mean(axis=(-1, -2))     # :658-662
    z = rel[..., 2]
    f[..., 7] = z.std(-1, ddof=1)
    f[..., 8] = z.max(-1) - z.min(-1)
    f[..., 9:12] = mean[:, :, 0, :]
    f[..., 12] = np.sqrt((rel.std(axis=2, ddof=1) ** 2).sum(-1))                 # :680
    return f, rel
