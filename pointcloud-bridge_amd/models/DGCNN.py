"""Drop-in for Highway_bridge/models/DGCNN.py of UT-Team-Chun/Pointcloud-bridge.

Same class name, constructor, methods (knn, get_graph_feature, forward), tensor layouts and
state_dict keys (including the doubled BatchNorm entries bn1.* / conv1.1.* that come from the
reference reusing self.bnX inside self.convX, DGCNN.py:12-33).  The dynamic graph (kNN in feature
space, rebuilt before every EdgeConv block) and the edge-feature gather run as gfx950 kernels
(..ops -> libpcb_hip.so); the [B,N,N] distance matrix of the reference never exists.

Activations are channels-last ([rows, C]) internally; the EdgeConv 1x1 convolutions are row GEMMs
over the parameters of the stock nn.Conv2d / nn.BatchNorm2d sub-modules.  GPU only.
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, rowmlp


class DGCNN(nn.Module):
    def __init__(self, num_classes=5, k=20):
        super().__init__()
        self.k = k
        self.bn1 = nn.BatchNorm2d(64)
        self.bn2 = nn.BatchNorm2d(64)
        self.bn3 = nn.BatchNorm2d(64)
        self.bn4 = nn.BatchNorm2d(128)
        self.bn5 = nn.BatchNorm1d(1024)

        def edge_block(cin, cout, bn):
            return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False), bn,
                                 nn.LeakyReLU(negative_slope=0.2))

        self.conv1 = edge_block(6, 64, self.bn1)
        self.conv2 = edge_block(64 * 2, 64, self.bn2)
        self.conv3 = edge_block(64 * 2, 64, self.bn3)
        self.conv4 = edge_block(64 * 2, 128, self.bn4)
        self.conv5 = nn.Sequential(nn.Conv1d(320, 1024, kernel_size=1, bias=False), self.bn5,
                                   nn.LeakyReLU(negative_slope=0.2))
        self.local_bn = nn.BatchNorm1d(320)
        self.point_conv = nn.Sequential(
            nn.Conv1d(1344, 512, 1), nn.BatchNorm1d(512), nn.LeakyReLU(negative_slope=0.2),
            nn.Conv1d(512, 256, 1), nn.BatchNorm1d(256), nn.LeakyReLU(negative_slope=0.2),
            nn.Conv1d(256, num_classes, 1))

    # -- reference API ------------------------------------------------------------------------
    def knn(self, x, k):
        """x [B,D,N] -> idx [B,N,k] int64, nearest first (reference DGCNN.py:49-70)."""
        return ops.knn(x.transpose(2, 1).contiguous(), k)

    def get_graph_feature(self, x, k=20, idx=None):
        """x [B,D,N] -> cat(x_j - x_i, x_i) as [B,2D,N,k] (reference DGCNN.py:72-109)."""
        xt = x.transpose(2, 1).contiguous()
        if idx is None:
            idx = ops.knn(xt, k)
        return ops.edge_features(xt, idx).permute(0, 3, 1, 2)

    # -- the first graph depends on the coordinates only -----------------------------------------
    def prefetch(self, xyz):
        """Build the coordinate graph of the NEXT batch on a side stream while the current batch is in
        its backward pass (the three feature-space graphs cannot move: they depend on the weights).
        The indices land in one of two buffers this module owns, used alternately -- the backward pass
        of the batch in flight still reads the other one -- and forward() picks them up by tensor identity."""
        from .pointnet2_utils import side_stream
        B, N, _ = xyz.shape
        k = min(self.k, N - 1)
        main = torch.cuda.current_stream()
        side = side_stream(xyz.device)
        side.wait_stream(main)
        self._parity = getattr(self, "_parity", 0) ^ 1
        if not hasattr(self, "_owned"):
            self._owned = {}
        with torch.cuda.stream(side):
            idx = ops.knn(xyz[:, :, :3].contiguous().float(), k)
            held = self._owned.get(self._parity)
            if held is None or held.shape != idx.shape:
                with torch.cuda.stream(main):
                    held = self._owned[self._parity] = torch.empty_like(idx)
            held.copy_(idx)
            ev = torch.cuda.Event()
            ev.record(side)
        # the weak reference pins the result to THIS tensor object: a later tensor that inherits the
        # address of a dropped batch must not pick up its graph
        self._graph0 = ((xyz.data_ptr(), xyz._version, tuple(xyz.shape), k), held, ev, weakref.ref(xyz))

    def _first_graph(self, xyz, x0, k):
        hit, self._graph0 = getattr(self, "_graph0", None), None
        if hit is not None and hit[3]() is xyz and hit[0] == (xyz.data_ptr(), xyz._version, tuple(xyz.shape), k):
            torch.cuda.current_stream().wait_event(hit[2])
            return hit[1]
        return None

    # -- forward ------------------------------------------------------------------------------
    def _edge_conv(self, block, x, k, idx=None):
        """x [B,N,D] channels-last -> [B,N,Cout]: kNN graph, edge features, conv+BN+LeakyReLU, max."""
        B, N, D = x.shape
        xf = x.float()
        if idx is None:
            idx = ops.knn(xf.contiguous(), k)  # the graph is always built from fp32 distances
        if ((x.requires_grad or not torch.is_grad_enabled()) and D >= 32 and D % 8 == 0
                and rowmlp.gathered_ok([block[0]], [block[1]])):
            # W [x_j - x_i ; x_i] = Wa x_j + (Wb - Wa) x_i: per-point products, gathered by the graph
            # (rowmlp.gathered_mlp); the [B*N*k, 2D] edge tensor is never written
            w = block[0].weight.view(block[0].out_channels, 2 * D)
            wa, wb = rowmlp.split_cols(w, D)
            xr = x.reshape(B * N, D)
            y = rowmlp.gathered_mlp([block[0]], [block[1]], rowmlp.point_linear(xr, wa),
                                    rowmlp.point_linear(xr, wb - wa), idx, rowmlp.ACT_LEAKY, pool=k)
            return y.view(B, N, -1)
        e = ops.edge_features(xf, idx).view(B * N * k, 2 * D)
        y = rowmlp.conv_bn_act(block[0], block[1], e, rowmlp.ACT_LEAKY, pool=k)
        return y.view(B, N, -1)

    def forward(self, xyz, features=None):
        """xyz [B,N,3], features [B,N,C] (ignored, as in the reference :123-128) -> logits [B,N,classes]."""
        B, N, _ = xyz.shape
        x0 = xyz[:, :, :3].contiguous()
        k = min(self.k, N - 1)  # reference :131
        x1 = self._edge_conv(self.conv1, x0, k, self._first_graph(xyz, x0, k))
        x2 = self._edge_conv(self.conv2, x1, k)
        x3 = self._edge_conv(self.conv3, x2, k)
        x4 = self._edge_conv(self.conv4, x3, k)
        local = torch.cat((x1, x2, x3, x4), dim=2).view(B * N, 320)
        local_norm = rowmlp.bn_act_rows(self.local_bn, local, rowmlp.ACT_LEAKY)
        g = rowmlp.conv_bn_act(self.conv5[0], self.conv5[1], local, rowmlp.ACT_LEAKY)
        g = rowmlp.scene_max(g, B, N)                         # adaptive_max_pool1d(x, 1), :160
        pf = rowmlp.scene_concat(local_norm, g, B, N)        # [local | pooled], :163-164  -> [B*N, 1344]
        pc = self.point_conv
        pf = rowmlp.conv_bn_act(pc[0], pc[1], pf, rowmlp.ACT_LEAKY)
        pf = rowmlp.conv_bn_act(pc[3], pc[4], pf, rowmlp.ACT_LEAKY)
        return rowmlp.conv_rows(pc[6], pf, torch.float32).view(B, N, -1)
