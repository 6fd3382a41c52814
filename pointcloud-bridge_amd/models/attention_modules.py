"""Drop-in for the bridge-specific encoders of Highway_bridge/models/attention_modules.py that the
reference's BridgeSeg network (EnhancedPointNet2, models/model.py:58-147) puts around the SA/FP trunk:
BridgeStructureEncoding (:523-687), GeometricFeatureExtraction (:241-269), ColorFeatureExtraction
(:690-753) and CompositeFeatureFusion (:756-772).  Same class names, constructor signatures, forward
layouts and state_dict keys.

What changes underneath:
  * the neighbour search is `pcb_knn` on the coordinates instead of torch.cdist + topk -- no
    [B,N,N] matrix (17 GB at B=16, N=16384).  The neighbours enter only through symmetric functions
    (moments, extrema, a max over k), so their order does not matter; the neighbour SET equals the
    reference's except where two candidates tie for the k-th place to within fp32 rounding of the
    two distance formulas (cdist takes sqrt(|a|^2+|b|^2-2ab) from one K=5 GEMM);
  * the gather, the 3x3 eigen-decompositions and the dozen reductions of get_structure_features are
    one kernel (`pcb_structure_features`);
  * the first 1x1 convolution of structure_mlp is split by input block: the 37 per-point channels
    (absolute encoding, structure descriptor) are multiplied once per point, only the 3 offset
    channels once per neighbour -- [B,N,k,40] is never built, and neither are the [B,C,N,k]
    activations: BatchNorm, ReLU, the second convolution and the max over k run inside
    ..nbrmlp (csrc/nbrmlp.hip) with the rows in registers;
  * ColorFeatureExtraction's neighbour search is dropped: the reference computes it and never uses the
    result (:736-743), so the output is identical without it.
The narrow (3..40 channel) layers stay fp32 in both precision modes: their 1x1 convolutions are
..rowsf32 (csrc/tinylin.hip; as GEMMs they have 3..16 columns and the library kernels picked for
them take 0.2-0.6 ms each), BatchNorm / ReLU / Sigmoid on them are ATen row ops.  The wide pointwise
layers of GeometricFeatureExtraction go through ..rowmlp like the SA/FP stacks.  GPU only.
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import nbrmlp, ops, rowmlp, rowsf32
from .pointnet2_utils import (_channels_last, _seq_rows, index_points, side_stream,  # noqa: F401
                              square_distance)  # the reference keeps its own copies of these two (:273-309)


def _rows_linear(x, w, b):
    """x [R,Ci] . w[Co,Ci]^T + b: the narrow-layer kernels for many rows, ATen for a handful (the
    per-scene context vector of ColorFeatureExtraction)."""
    if x.shape[0] >= 1024 and max(w.shape) <= rowsf32.MAX_CHANNELS:
        return rowsf32.rows_linear(x, w, b)
    return F.linear(x, w, b)


def _rows_conv(conv, x):
    return _rows_linear(x, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)


def _rows_seq_f32(seq, x):
    """An nn.Sequential of 1x1 Conv / BatchNorm / ReLU / Sigmoid on fp32 rows [R, C].  BatchNorm (with the ReLU behind
    it) runs on the narrow-row kernels (rowsf32.bn_act_rows: any channel count, kernels only -- a captured step may
    contain it); SyncBatchNorm and a handful of rows stay with ATen."""
    seq = list(seq)
    i = 0
    while i < len(seq):
        m = seq[i]
        i += 1
        if isinstance(m, (nn.Conv1d, nn.Conv2d)):
            x = _rows_conv(m, x)
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)) and not isinstance(m, nn.SyncBatchNorm) \
                and x.is_cuda and x.shape[0] >= 1024 and x.shape[1] <= rowsf32.MAX_CHANNELS:
            relu = i < len(seq) and isinstance(seq[i], nn.ReLU)
            x = rowsf32.bn_act_rows(m, x, rowsf32.ACT_RELU if relu else rowsf32.ACT_NONE)
            i += int(relu)
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.SyncBatchNorm)):
            x = rowmlp._bn_rows_fp32(m, x)
        elif isinstance(m, nn.ReLU):
            x = F.relu(x)
        elif isinstance(m, nn.Sigmoid):
            x = torch.sigmoid(x)
        else:
            raise TypeError(f"unsupported layer in pointwise stack: {type(m).__name__}")
    return x


_geometry = {}  # parked results of BridgeStructureEncoding.prefetch, keyed by tensor identity


class BridgeStructureEncoding(nn.Module):
    """Absolute grid encoding + k-neighbourhood structure descriptor -> per-point code.
    forward(xyz [B,N,3]) -> [B, channels, N]  (reference :577-618)."""

    def __init__(self, channels=32, k_neighbors=16, freq_bands=4, min_scale=0.05, max_scale=100.0,
                 grid_size=1.0):
        super().__init__()
        self.channels = channels
        self.k = k_neighbors
        self.freq_bands = freq_bands
        self.min_scale = min_scale
        self.max_scale = max_scale
        self.grid_size = grid_size
        self.register_buffer('freqs', 2.0 ** torch.linspace(0., freq_bands - 1, freq_bands))
        self.abs_pos_dim = 6 * freq_bands
        self.rel_pos_dim = 3
        self.local_struct_dim = 13
        self.total_dim = self.abs_pos_dim + self.rel_pos_dim + self.local_struct_dim
        self.structure_mlp = nn.Sequential(
            nn.Conv2d(self.total_dim, channels, 1),
            nn.BatchNorm2d(channels),
            nn.ReLU(),
            nn.Conv2d(channels, channels, 1))

    def compute_absolute_position_encoding(self, xyz):
        """sin/cos of the grid-snapped coordinates at every frequency, [B,N,6*freq_bands] (:552-575)."""
        grid = torch.floor(xyz / self.grid_size) * self.grid_size
        scaled = grid.unsqueeze(-2) * self.freqs.view(-1, 1)               # [B,N,F,3]
        enc = torch.stack([torch.sin(scaled), torch.cos(scaled)], dim=-2)  # [B,N,F,2,3]
        return enc.flatten(-3)

    def get_structure_features(self, rel_pos):
        """Reference API (:620): rel_pos [B,N,k,3] -> [B,N,13].  The descriptor kernel works from
        coordinates + indices; offsets alone are served by treating every neighbourhood as its own
        tiny cloud (centre at the origin followed by the k offsets)."""
        B, N, k, _ = rel_pos.shape
        cloud = torch.cat([rel_pos.new_zeros(B * N, 1, 3), rel_pos.reshape(B * N, k, 3)], dim=1)
        idx = torch.arange(1, k + 1, device=rel_pos.device).expand(B * N, k + 1, k).contiguous()
        feat, _ = ops.structure_features(cloud.contiguous(), idx, with_offsets=False)
        return feat[:, 0].reshape(B, N, 13)

    def geometry(self, xyz):
        """Everything that depends on the coordinates only: per-point channels [B,N,6F+13] (absolute
        encoding, structure descriptor) and the neighbour offsets rel [B,N,k,3].  No gradient."""
        k = min(self.k, xyz.shape[1])
        with torch.no_grad():
            idx = ops.knn(xyz, k)                                       # :584-586
            struct, rel = ops.structure_features(xyz, idx)             # :595-603
            per_point = torch.cat([self.compute_absolute_position_encoding(xyz), struct], dim=-1)
        return per_point, rel

    def _geometry_key(self, xyz):
        return (xyz.data_ptr(), xyz._version, tuple(xyz.shape), min(self.k, xyz.shape[1]), self.freq_bands,
                float(self.grid_size))

    def prefetch(self, xyz):
        """Compute geometry(xyz) of the NEXT batch on the side stream that also carries the FPS
        pyramid (pointnet2_utils.prefetch_sampling), while the current batch is in its backward
        pass; rows() picks it up by tensor identity.  Same results, only earlier."""
        xyz = xyz.float().contiguous()
        main = torch.cuda.current_stream()
        side = side_stream(xyz.device)
        side.wait_stream(main)
        _geometry.clear()
        # The results (rel alone is 100 MB at B=16, N=16384, k=32) are copied into two alternating sets
        # of buffers this module owns -- the backward pass of the batch in flight still reads the
        # other set -- so that no large block changes streams in the caching allocator (see
        # pointnet2_utils.prefetch_sampling for the measurement behind this).
        self._parity = getattr(self, "_parity", 0) ^ 1
        with torch.cuda.stream(side):
            res = self.geometry(xyz)
            held = self._owned.get(self._parity) if hasattr(self, "_owned") else None
            if held is None or any(h.shape != r.shape for h, r in zip(held, res)):
                with torch.cuda.stream(main):
                    held = tuple(torch.empty_like(r) for r in res)
                if not hasattr(self, "_owned"):
                    self._owned = {}
                self._owned[self._parity] = held
            for h, r in zip(held, res):
                h.copy_(r)
            ev = torch.cuda.Event()
            ev.record(side)
        # pinned to THIS tensor object by a weak reference: a later tensor that inherits the address of
        # a dropped batch must not pick up its neighbourhoods
        _geometry[self._geometry_key(xyz)] = (held[0], held[1], ev, weakref.ref(xyz))

    def rows(self, xyz):
        """xyz [B,N,3] -> code rows [B*N, channels] (channels-last)."""
        B, N, _ = xyz.shape
        xyz = xyz.float().contiguous()
        k = min(self.k, N)
        hit = _geometry.pop(self._geometry_key(xyz), None)
        if hit is not None and hit[3]() is not xyz:
            hit = None
        from . import pointnet2_utils as pu
        static = pu.static_sampling()
        job = static.lookup_job(("geometry", id(self)), xyz) if static is not None else None
        if job is not None:
            per_point, rel = job         # a captured step: computed beside the previous step's backward pass (StaticSampling)
        elif hit is None:
            per_point, rel = self.geometry(xyz)
        else:
            per_point, rel, ev, _ = hit
            torch.cuda.current_stream().wait_event(ev)
        conv0, bn, _, conv1 = self.structure_mlp
        a = self.abs_pos_dim
        w = conv0.weight.view(self.channels, self.total_dim)
        # conv0 over cat(abs, rel, struct) (:606-614) = per-point part + per-neighbour part
        base = _rows_linear(per_point.view(B * N, -1), torch.cat([w[:, :a], w[:, a + 3:]], dim=1), conv0.bias)
        if isinstance(bn, nn.SyncBatchNorm) or self.channels > 16 or k > 255:
            # statistics shared across ranks / widths beyond the fused kernel: the same algebra as
            # torch ops on the materialised [P*k, C] rows
            y = (rel.view(B * N, k, 3) @ w[:, a:a + 3].t()) + base.unsqueeze(1)
            y = F.relu(rowmlp._bn_rows_fp32(bn, y.view(B * N * k, self.channels)))
            return _rows_conv(conv1, y).view(B * N, k, self.channels).max(dim=1)[0]
        return nbrmlp.neighbour_mlp(base, rel.view(B * N, k, 3), w[:, a:a + 3], bn, conv1)  # :614-616

    def forward(self, xyz):
        B, N, _ = xyz.shape
        return self.rows(xyz).view(B, N, self.channels).transpose(1, 2)


class GeometricFeatureExtraction(nn.Module):
    """Features + 16-channel structure code -> pointwise MLP (reference :241-269).
    forward(x [B,C,N], xyz [B,N,3]) -> [B,C,N]."""

    def __init__(self, in_channels):
        super().__init__()
        self.mlp = nn.Sequential(
            nn.Conv1d(in_channels + 16, in_channels, 1),
            nn.BatchNorm1d(in_channels),
            nn.ReLU(),
            nn.Conv1d(in_channels, in_channels, 1))
        self.br_pos = BridgeStructureEncoding(channels=16)

    def forward(self, x, xyz):
        B, N, _ = xyz.shape
        pos = self.br_pos.rows(xyz)
        rows = _channels_last(x).reshape(B * N, -1)
        rows = torch.cat([rows, pos.to(rows.dtype)], dim=1)              # :262-265
        return _seq_rows(self.mlp, rows).view(B, N, -1).transpose(1, 2)


class ColorFeatureExtraction(nn.Module):
    """Pointwise colour code with channel attention and a global context gate (reference :690-753).
    forward(colors [B,3,N], xyz [B,N,3]) -> [B,out_channels,N]; xyz is accepted and, as in the
    reference, has no influence on the result."""

    def __init__(self, in_channels=3, out_channels=32):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.color_mlp = nn.Sequential(
            nn.Conv1d(in_channels, 16, 1), nn.BatchNorm1d(16), nn.ReLU(),
            nn.Conv1d(16, out_channels, 1), nn.BatchNorm1d(out_channels), nn.ReLU())
        self.color_attention = nn.Sequential(
            nn.Conv1d(out_channels, out_channels, 1), nn.BatchNorm1d(out_channels), nn.ReLU(),
            nn.Conv1d(out_channels, out_channels, 1), nn.Sigmoid())
        self.color_context = nn.Sequential(
            nn.AdaptiveAvgPool1d(1),
            nn.Conv1d(out_channels, out_channels // 2, 1), nn.ReLU(),
            nn.Conv1d(out_channels // 2, out_channels, 1), nn.Sigmoid())

    def rows(self, colors_rows, B, N):
        """colors [B*N, 3] fp32 -> [B*N, out_channels]."""
        feat = _rows_seq_f32(self.color_mlp, colors_rows)                        # :724
        local = feat * _rows_seq_f32(self.color_attention, feat)                 # :746-747
        ctx = _rows_seq_f32(list(self.color_context)[1:], rowsf32.scene_mean(feat, B, N))    # :750
        return rowsf32.scene_scale(local, ctx, B, N)                             # :751

    def forward(self, colors, xyz=None):
        B, _, N = colors.shape
        rows = colors.float().transpose(1, 2).reshape(B * N, -1)
        return self.rows(rows, B, N).view(B, N, -1).transpose(1, 2)


class CompositeFeatureFusion(nn.Module):
    """cat(spatial, colour) -> Conv1d -> BN -> ReLU (reference :756-772)."""

    def __init__(self, spatial_channels, color_channels):
        super().__init__()
        total_channels = spatial_channels + color_channels
        self.fusion_mlp = nn.Sequential(
            nn.Conv1d(total_channels, spatial_channels, 1),
            nn.BatchNorm1d(spatial_channels),
            nn.ReLU())

    def rows(self, spatial_rows, color_rows):
        return _rows_seq_f32(self.fusion_mlp, torch.cat([spatial_rows, color_rows], dim=1))

    def forward(self, spatial_features, color_features):
        B, _, N = spatial_features.shape
        s = spatial_features.float().transpose(1, 2).reshape(B * N, -1)
        c = color_features.float().transpose(1, 2).reshape(B * N, -1)
        return self.rows(s, c).view(B, N, -1).transpose(1, 2)
