"""Segmentation networks assembled from the drop-in operators.

The reference's container files (Highway_bridge/models/model.py, models/pointnet2.py) only wire
SetAbstraction / FeaturePropagation modules together and work unchanged on top of
`pointnet2_utils`.  They do not travel to the GPU box, so the benchmark, smoke test and parity
tests use these equivalents; attribute names (sa1..3, fp3..1, conv1, bn1, drop1, conv2, fusion,
final_fusion) and therefore state_dict keys match the reference classes named in each docstring.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import rowmlp
from .attention_modules import (BridgeStructureEncoding, ColorFeatureExtraction, CompositeFeatureFusion,
                                GeometricFeatureExtraction)
from .pointnet2_utils import (EnhancedFeaturePropagation, FeaturePropagation, MultiScaleSetAbstraction,
                              SetAbstraction, _channels_last, _seq_rows, prefetch_sampling, run_branches)


class _SamplingPrefetchMixin:
    _next_xyz = None
    # The seam between encoder and decoder: a callable (tensors...) -> tensors applied to everything the encoder hands to
    # the decoder, or None.  A step captured as TWO hipGraphs (bench.py, parallel: the decoder's gradient bucket travels
    # while the encoder's backward pass still runs) installs one that detaches the tensors and records both sides, so that
    # autograd can be run down to the seam and, separately, from the seam to the inputs.
    decoder_cut = None
    # (first top-level child module of the decoder: the parameters in front of it form the encoder's gradient bucket)
    decoder_first = "fp3"

    @property
    def sampling(self):
        """This model's pointnet2_utils.SamplingState: its prefetched pyramid, parked neighbour tables, installed static
        pipeline and scene shard.  Current for the duration of every call of the model (see __call__)."""
        st = self.__dict__.get("_sampling_state")
        if st is None:
            from .pointnet2_utils import SamplingState
            st = self.__dict__["_sampling_state"] = SamplingState()
        return st

    def __call__(self, *args, **kwargs):
        from .pointnet2_utils import sampling_scope
        with sampling_scope(self.sampling):
            return super().__call__(*args, **kwargs)

    def _cut(self, *tensors):
        return tensors if self.decoder_cut is None else tuple(self.decoder_cut(*tensors))

    def prefetch(self, xyz):
        """Start the FPS pyramid of the NEXT batch's coordinates on a side stream (see
        pointnet2_utils.prefetch_sampling); call between forward and backward of the current batch."""
        sas = (self.sa1, self.sa2, self.sa3)
        balls = [((m.radius_list, m.nsample_list) if hasattr(m, "radius_list") else ([m.radius], [m.nsample]))
                 for m in sas]
        k = 4 if hasattr(self.fp1, "attention") else 3   # EnhancedFeaturePropagation uses 4 neighbours (:256)
        # decoder stages: fp3 level 2 <- 3, fp2 level 1 <- 2, fp1 level 0 <- 1
        from .pointnet2_utils import sampling_scope
        with sampling_scope(self.sampling):
            prefetch_sampling(xyz.contiguous(), [m.npoint for m in sas], balls, [(2, 3, k), (1, 2, k), (0, 1, k)])

    def static_sampling(self, xyz):
        """A pointnet2_utils.StaticSampling pipeline for steps replayed from a hipGraph: the same coordinate-only
        work `prefetch` issues (FPS pyramid, ball queries, decoder k-NN and inverted indices), in persistent
        double-set buffers."""
        from .pointnet2_utils import StaticSampling
        sas = (self.sa1, self.sa2, self.sa3)
        balls = [((m.radius_list, m.nsample_list) if hasattr(m, "radius_list") else ([m.radius], [m.nsample]))
                 for m in sas]
        k = 4 if hasattr(self.fp1, "attention") else 3
        from .pointnet2_utils import sampling_scope
        with sampling_scope(self.sampling):
            return StaticSampling(xyz, [m.npoint for m in sas], balls, [(2, 3, k), (1, 2, k), (0, 1, k)])

    def set_next(self, xyz):
        """Pipelined inference: the coordinates of the batch that FOLLOWS the next forward call.  That
        call starts their coordinate-only work (prefetch) as soon as its own encoder has consumed the
        previous prefetch, so that it runs beside the decoder instead of in front of the next pass."""
        self._next_xyz = xyz

    def _start_next(self):
        if self._next_xyz is not None:
            nxt, self._next_xyz = self._next_xyz, None
            from . import pointnet2_utils as pu
            if pu.static_sampling() is not None:      # a captured pass: into the pipeline's staging set, on its side stream
                pu.static_sampling().compute_beside(nxt.contiguous())
            else:
                self.prefetch(nxt)


# (npoint, radius, nsample, in_channel, mlp) -- models/model.py:17-19 == models/pointnet2.py:20-22
_SSG_ENCODER = [
    (1024, 0.1, 32, 6, [64, 64, 128]),
    (256, 0.2, 32, 131, [128, 128, 256]),
    (64, 0.4, 32, 259, [256, 256, 512]),
]
# (npoint, radii, nsamples, in_channel, mlp) -- models/model.py:73-76
_MSG_ENCODER = [
    (1024, [0.1, 0.2], [16, 32], 6, [64, 64, 128]),
    (512, [0.2, 0.4], [16, 32], 259, [128, 128, 256]),
    (128, [0.4, 0.8], [16, 32], 515, [256, 256, 512]),
]


class PointNet2(_SamplingPrefetchMixin, nn.Module):
    """PointNet++ SSG segmentation net.

    rgb_skip=False: `PointNet2` of models/model.py:12-56 (fp1 sees only the propagated features).
    rgb_skip=True:  `PointNet2` of models/pointnet2.py:10-61 (fp1 also sees the raw colours, 131 ch).
    forward(xyz [B,N,3], points [B,N,3]) -> logits [B,num_classes,N].
    """

    def __init__(self, num_classes=8, rgb_skip=False, encoder=None):
        super().__init__()
        self.rgb_skip = rgb_skip
        enc = encoder or _SSG_ENCODER
        self.sa1 = SetAbstraction(*enc[0])
        self.sa2 = SetAbstraction(*enc[1])
        self.sa3 = SetAbstraction(*enc[2])
        self.fp3 = FeaturePropagation(768, [256, 256])
        self.fp2 = FeaturePropagation(384, [256, 128])
        self.fp1 = FeaturePropagation(128 + (3 if rgb_skip else 0), [128, 128, 128])
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_classes, 1)

    def forward(self, xyz, points):
        points = points.transpose(1, 2)
        l1_xyz, l1 = self.sa1(xyz, points)
        l2_xyz, l2 = self.sa2(l1_xyz, l1)
        l3_xyz, l3 = self.sa3(l2_xyz, l2)
        self._start_next()
        l1, l2, l3 = self._cut(l1, l2, l3)
        l2 = self.fp3(l2_xyz, l3_xyz, l2, l3)
        l1 = self.fp2(l1_xyz, l2_xyz, l1, l2)
        l0 = self.fp1(xyz, l1_xyz, points if self.rgb_skip else None, l1)
        B, _, N = l0.shape
        feat = rowmlp.conv_bn_act(self.conv1, self.bn1, _channels_last(l0).view(B * N, -1))
        logits = rowmlp.conv_rows(self.conv2, rowmlp.dropout_rows(self.drop1, feat), torch.float32)
        return logits.view(B, N, -1).transpose(1, 2)


class MultiScaleFeatureFusion(nn.Module):
    """models/model.py:149-167: resample every decoder level to N points, 1x1 conv each, concatenate."""

    coarse_rows = True  # run each level's layer before the upsampling (see forward)

    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        self.convs = nn.ModuleList(
            nn.Sequential(nn.Conv1d(c, out_channels, 1), nn.BatchNorm1d(out_channels), nn.ReLU())
            for c in in_channels_list)

    def forward(self, features_list):
        """[B,C_i,S_i] levels -> [B, N, 3*out] channels-last rows (nearest resampling along the
        point axis, exactly F.interpolate(feat, size=N) of models/model.py:164)."""
        return self.concat(*self.levels(features_list))

    @staticmethod
    def concat(outs, reps, B, n):
        """[B, N, sum C] from levels(): every level repeated to N rows per scene, side by side."""
        q = rowmlp.mode().q
        if all(r == 1 for r in reps) or any(o.shape[1] % q for o in outs) or len(outs) > 4:
            return torch.cat([o.view(B, n // r, 1, -1).expand(B, n // r, r, o.shape[1]).reshape(B, n, -1)
                              for o, r in zip(outs, reps)], dim=2)
        # one pass writes every level, broadcast over its repeats, into its column block; backward hands every
        # level its block of the gradient summed over the repeats (rowmlp.repeat_concat)
        return rowmlp.repeat_concat(outs, reps).view(B, n, -1)

    def levels(self, features_list):
        """The levels after their layers, BEFORE upsampling and concatenation: (rows [B*N/r_i, out] per level, how
        often each row is repeated, B, N).  forward() concatenates them; a consumer that starts with a pointwise conv
        can take them apart instead (rowmlp.conv_bn_act_levels)."""
        n = features_list[2].shape[2]
        B = features_list[0].shape[0]

        def level(f, conv):
            """(rows of this level after its layer, how often each is repeated)"""
            _, _, S = f.shape
            rows = _channels_last(f)
            r = n // S
            if S != n and n % S == 0 and (r & (r - 1)) == 0:
                # scale = S/n = 2^-j is exact in fp32: nearest source of point i is i // r, i.e.
                # every coarse row repeated r times (its backward is a plain sum over r rows)
                if self.coarse_rows and conv[0].out_channels % rowmlp.mode().q == 0:
                    # A pointwise layer commutes with the repetition, and batch statistics over
                    # rows repeated r times each equal those over the distinct rows: run the layer
                    # on the S coarse rows (r times less GEMM work, forward and backward) and
                    # repeat its output.  Only the sample count of the unbiased running variance
                    # differs, which stat_repeat restores.
                    return rowmlp.conv_bn_act(conv[0], conv[1], rows.reshape(B * S, -1), rowmlp.ACT_RELU, stat_repeat=r), r
                rows = rows.unsqueeze(2).expand(B, S, r, rows.shape[2]).reshape(B, n, rows.shape[2])
            elif S != n:
                ramp = torch.arange(S, dtype=torch.float32, device=f.device).view(1, 1, S)
                src = F.interpolate(ramp, size=n).view(n).long()  # the very index map F.interpolate uses
                rows = rows.index_select(1, src)
            return _seq_rows(conv, rows.reshape(B * n, -1)), 1

        # the levels are independent: the full-resolution one on the caller's stream, the coarse ones beside it
        order = sorted(range(len(features_list)), key=lambda i: -features_list[i].shape[2])
        reps_of = {}

        def run(i):
            o, r = level(features_list[i], self.convs[i])
            reps_of[i] = r
            return o

        res = run_branches(features_list[0].device, [(lambda i=i: run(i)) for i in order], inputs=tuple(features_list))
        outs = [None] * len(order)
        for i, o in zip(order, res):
            outs[i] = o
        reps = [reps_of[i] for i in range(len(order))]
        return outs, reps, B, n


class PointNet2MSG(_SamplingPrefetchMixin, nn.Module):
    """PointNet++ MSG segmentation net = the SA/FP trunk of `EnhancedPointNet2` (BridgeSeg),
    models/model.py:58-147: MSG encoder (:73-76), EnhancedFeaturePropagation decoder (:84-86),
    MultiScaleFeatureFusion (:88-91) and the final_fusion head (:93-99).

    The bridge-specific encoders in front of the trunk (BridgeStructureEncoding,
    ColorFeatureExtraction, CompositeFeatureFusion, GeometricFeatureExtraction --
    models/attention_modules.py) are outside this path; the colours
    enter sa1 directly, which keeps its reference width of 6 channels.
    forward(xyz [B,N,3], features [B,N,3]) -> logits [B,num_classes,N].
    """

    def __init__(self, num_classes=5, encoder=None):
        super().__init__()
        enc = encoder or _MSG_ENCODER
        self.sa1 = MultiScaleSetAbstraction(*enc[0])
        self.sa2 = MultiScaleSetAbstraction(*enc[1])
        self.sa3 = MultiScaleSetAbstraction(*enc[2])
        self.fp3 = EnhancedFeaturePropagation(1536, [1024, 256])
        self.fp2 = EnhancedFeaturePropagation(512, [256, 256])
        self.fp1 = EnhancedFeaturePropagation(256 + 3, [256, 128])
        self.fusion = MultiScaleFeatureFusion([256, 256, 128], 128)
        self.final_fusion = nn.Sequential(
            nn.Conv1d(384, 128, 1), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.5),
            nn.Conv1d(128, num_classes, 1))

    def forward(self, xyz, features):
        feats = features.transpose(1, 2)
        l1_xyz, l1 = self.sa1(xyz, feats)
        l2_xyz, l2 = self.sa2(l1_xyz, l1)
        l3_xyz, l3 = self.sa3(l2_xyz, l2)
        self._start_next()
        l1, l2, l3 = self._cut(l1, l2, l3)
        l2 = self.fp3(l2_xyz, l3_xyz, l2, l3)
        l1 = self.fp2(l1_xyz, l2_xyz, l1, l2)
        l0 = self.fp1(xyz, l1_xyz, feats, l1)
        # fusion's upsample + concatenate [B,N,384] and final_fusion's first conv as one layer over the three levels
        outs, reps, B, N = self.fusion.levels([l2, l1, l0])
        ff = self.final_fusion
        x = rowmlp.conv_bn_act_levels(ff[0], ff[1], outs, reps,
                                      concat=lambda: self.fusion.concat(outs, reps, B, N).view(B * N, -1))
        logits = rowmlp.conv_rows(ff[4], rowmlp.dropout_rows(ff[3], x), torch.float32)
        return logits.view(B, N, -1).transpose(1, 2)


class EnhancedPointNet2(_SamplingPrefetchMixin, nn.Module):
    """The reference's BridgeSeg network, models/model.py:58-147, whole: structure and colour encoders
    in front (bri_enc, color_encoder, feature_fusion), MSG encoder, GeometricFeatureExtraction after
    sa2 / sa3 (geometric1 is constructed and, as in the reference :130, not called), EFP decoder,
    multi-scale fusion, final_fusion head.  `cls_head` exists only for its state_dict keys (:100-111).
    forward(xyz [B,N,3], features [B,N,3]) -> logits [B,num_classes,N].
    """

    def __init__(self, num_classes=5):
        super().__init__()
        input_ch = 3
        self.bri_enc = BridgeStructureEncoding(input_ch, 32, 4)
        self.color_encoder = ColorFeatureExtraction(3, 6)
        self.feature_fusion = CompositeFeatureFusion(input_ch, 6)
        self.sa1 = MultiScaleSetAbstraction(*_MSG_ENCODER[0])
        self.sa2 = MultiScaleSetAbstraction(*_MSG_ENCODER[1])
        self.sa3 = MultiScaleSetAbstraction(*_MSG_ENCODER[2])
        self.geometric1 = GeometricFeatureExtraction(128 * 2)
        self.geometric2 = GeometricFeatureExtraction(256 * 2)
        self.geometric3 = GeometricFeatureExtraction(512 * 2)
        self.fp3 = EnhancedFeaturePropagation(1536, [1024, 256])
        self.fp2 = EnhancedFeaturePropagation(512, [256, 256])
        self.fp1 = EnhancedFeaturePropagation(256 + input_ch, [256, 128])
        self.fusion = MultiScaleFeatureFusion([256, 256, 128], 128)
        self.final_fusion = nn.Sequential(
            nn.Conv1d(384, 128, 1), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.5),
            nn.Conv1d(128, num_classes, 1))
        self.num_classes = num_classes
        self.cls_head = nn.Sequential(
            nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Dropout(0.5),
            nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
            nn.Linear(256, num_classes))

    def prefetch(self, xyz):
        """Coordinate-only work of the NEXT batch on the side stream: the kNN graph + structure
        descriptor of bri_enc, then the FPS pyramid."""
        xyz = xyz.contiguous()
        self.bri_enc.prefetch(xyz)
        super().prefetch(xyz)

    def static_sampling(self, xyz):
        """The pipeline of a captured step: the trunk's (pyramid, ball queries, decoder k-NN) plus the neighbourhood
        geometry of the three BridgeStructureEncoding modules -- bri_enc on the input cloud, geometric2 / geometric3 on
        levels 2 and 3 -- which depend on coordinates only (models/attention_modules.py:584-603)."""
        static = super().static_sampling(xyz)
        for level, enc in ((0, self.bri_enc), (2, self.geometric2.br_pos), (3, self.geometric3.br_pos)):
            static.jobs.append((("geometry", id(enc)), level, enc.geometry))
        return static

    def forward(self, xyz, features=None):
        B, N, _ = xyz.shape
        pos = self.bri_enc.rows(xyz)                                              # :119
        col = self.color_encoder.rows(features.float().reshape(B * N, -1), B, N)  # :122
        fused_in = self.feature_fusion.rows(pos, col).view(B, N, -1).transpose(1, 2)  # :123
        l1_xyz, l1 = self.sa1(xyz, fused_in)
        l2_xyz, l2 = self.sa2(l1_xyz, l1)
        l2 = self.geometric2(l2, l2_xyz)                                          # :133
        l3_xyz, l3 = self.sa3(l2_xyz, l2)
        l3 = self.geometric3(l3, l3_xyz)                                          # :136
        self._start_next()
        l1, l2, l3, fused_dec = self._cut(l1, l2, l3, fused_in)
        l2 = self.fp3(l2_xyz, l3_xyz, l2, l3)
        l1 = self.fp2(l1_xyz, l2_xyz, l1, l2)
        l0 = self.fp1(xyz, l1_xyz, fused_dec, l1)
        # fusion's upsample + concatenate [B,N,384] and final_fusion's first conv as one layer over the three levels
        outs, reps, B, N = self.fusion.levels([l2, l1, l0])
        ff = self.final_fusion
        x = rowmlp.conv_bn_act_levels(ff[0], ff[1], outs, reps,
                                      concat=lambda: self.fusion.concat(outs, reps, B, N).view(B * N, -1))
        logits = rowmlp.conv_rows(ff[4], rowmlp.dropout_rows(ff[3], x), torch.float32)
        return logits.view(B, N, -1).transpose(1, 2)
