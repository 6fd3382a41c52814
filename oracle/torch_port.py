"""CPU port of the reference's hot path as plain PyTorch (ATen) compositions.

TEST / BASELINE INFRASTRUCTURE ONLY.  Used by (1) tests/ as a second witness next to the C oracle
and (2) bench.py's `cpu_baseline` leg ("kind": "port"): the reference's own files do not travel
to the GPU box, so its CPU path is restated here with the same ATen operations it is made of --
matmul-expansion distances, full sorts, a Python loop for FPS, advanced-indexing gathers, 1x1
convolutions + BatchNorm -- so that its cost profile on the host cores is the reference's.
The product package never imports this module.

Parity status: pinned -- tests/test_torch_port_cpu.py checks it against the golden vectors that
tests/golden/make_golden*.py generated from the reference itself.

The functions execute the *same parameter containers* as the product (the nn.Modules in
pointcloud-bridge_amd/models), moved to the CPU: `run(model, *inputs)` dispatches on the module
types.  Cited lines: /root/reference/Highway_bridge/models/pointnet2_utils.py and DGCNN.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- index operators
def pairwise_sqdist(src, dst):
    """pointnet2_utils.py:7-14: -2 * src @ dst^T, then += |src|^2, then += |dst|^2."""
    d = torch.matmul(src, dst.transpose(1, 2)).mul_(-2)
    d += (src * src).sum(-1).unsqueeze(2)
    d += (dst * dst).sum(-1).unsqueeze(1)
    return d


def take_rows(points, idx):
    """pointnet2_utils.py:17-39: batched gather, idx clamped to [0, N-1]."""
    B = points.shape[0]
    idx = idx.clamp(0, points.shape[1] - 1)
    b = torch.arange(B, device=points.device).view(B, *([1] * (idx.dim() - 1)))
    return points[b, idx]


def fps(xyz, npoint):
    """pointnet2_utils.py:63-80 (one torch.randint from the CPU generator, then the S-step loop)."""
    B, N, _ = xyz.shape
    out = torch.zeros(B, npoint, dtype=torch.long)
    running = torch.full((B, N), 1e10)
    far = torch.randint(0, N, (B,), dtype=torch.long)
    rows = torch.arange(B)
    for s in range(npoint):
        out[:, s] = far
        c = xyz[rows, far].unsqueeze(1)
        d = ((xyz - c) ** 2).sum(-1)
        running = torch.where(d < running, d, running)
        far = running.max(-1)[1]
    return out


def ball(radius, nsample, xyz, new_xyz):
    """pointnet2_utils.py:97-112: mask by radius, full sort of the index matrix, first nsample."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.arange(N).view(1, 1, N).repeat(B, S, 1)
    idx[pairwise_sqdist(new_xyz, xyz) > radius ** 2] = N
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[:, :, :1].expand(-1, -1, nsample)
    return torch.where(idx == N, first, idx)


def nearest_k(xyz1, xyz2, k):
    """pointnet2_utils.py:185-188 / :253-256: full sort of the [B,N,S] matrix, first k."""
    d, i = pairwise_sqdist(xyz1, xyz2).sort(dim=-1)
    return d[:, :, :k], i[:, :, :k]


def knn_graph(x_bdn, k):
    """DGCNN.py:49-70."""
    xt = x_bdn.transpose(2, 1).contiguous()
    inner = -2 * torch.matmul(xt, xt.transpose(2, 1))
    xx = (xt ** 2).sum(dim=2, keepdim=True)
    return (-(xx + inner + xx.transpose(2, 1))).topk(k=k, dim=-1)[1]


# ----------------------------------------------------------------------------- modules
def _stack2d(convs, bns, x):
    for conv, bn in zip(convs, bns):
        x = F.relu(bn(conv(x)))
    return x


def set_abstraction(mod, xyz, points):
    """SetAbstraction.forward, pointnet2_utils.py:131-156 (FPS, ball query, group, MLP, max)."""
    B = xyz.shape[0]
    S = mod.npoint
    new_xyz = take_rows(xyz, fps(xyz, S))
    idx = ball(mod.radius, mod.nsample, xyz, new_xyz)
    grouped = take_rows(xyz, idx) - new_xyz.view(B, S, 1, 3)
    if points is not None:
        grouped = torch.cat([grouped, take_rows(points.transpose(1, 2), idx)], dim=-1)
    x = _stack2d(mod.mlp_convs, mod.mlp_bns, grouped.permute(0, 3, 1, 2).contiguous())
    return new_xyz, x.max(-1)[0]


def msg_set_abstraction(mod, xyz, points):
    """MultiScaleSetAbstraction.forward, pointnet2_utils.py:326-360."""
    B = xyz.shape[0]
    S = mod.npoint
    new_xyz = take_rows(xyz, fps(xyz, S))
    outs = []
    for i, (r, ns) in enumerate(zip(mod.radius_list, mod.nsample_list)):
        idx = ball(r, ns, xyz, new_xyz)
        grouped = take_rows(xyz, idx) - new_xyz.view(B, S, 1, 3)
        if points is not None:
            grouped = torch.cat([grouped, take_rows(points.transpose(1, 2), idx)], dim=-1)
        x = _stack2d(mod.conv_blocks[i], mod.bn_blocks[i], grouped.permute(0, 3, 1, 2))
        outs.append(x.max(-1)[0])
    return new_xyz, torch.cat(outs, dim=1)


def _propagate(xyz1, xyz2, points1, points2, k):
    B, N, _ = xyz1.shape
    d, i = nearest_k(xyz1, xyz2, k)
    w = 1.0 / (d + 1e-8)
    w = w / w.sum(dim=2, keepdim=True)
    x = (take_rows(points2.transpose(1, 2), i) * w.view(B, N, k, 1)).sum(dim=2)
    if points1 is not None:
        x = torch.cat([points1.transpose(1, 2), x], dim=-1)
    return x.transpose(1, 2)


def feature_propagation(mod, xyz1, xyz2, points1, points2):
    """FeaturePropagation.forward, pointnet2_utils.py:171-211."""
    return _stack2d(mod.mlp_convs, mod.mlp_bns, _propagate(xyz1, xyz2, points1, points2, 3))


def enhanced_feature_propagation(mod, xyz1, xyz2, points1, points2):
    """EnhancedFeaturePropagation.forward, pointnet2_utils.py:246-298."""
    x = _propagate(xyz1, xyz2, points1, points2, 4)
    x = x * mod.attention(x)
    edge = mod.boundary_aware(xyz1.transpose(1, 2))
    y = _stack2d(mod.mlp_convs, mod.mlp_bns, x)
    if mod.skip_connection:
        y = y + x
    return y + edge


def _sa(mod, xyz, points):
    return (msg_set_abstraction if hasattr(mod, "conv_blocks") else set_abstraction)(mod, xyz, points)


def _fp(mod, *a):
    return (enhanced_feature_propagation if hasattr(mod, "attention") else feature_propagation)(mod, *a)


def pointnet2(model, xyz, colors):
    """PointNet2 (models/model.py:32-56, models/pointnet2.py:36-61) on the product's container."""
    pts = colors.transpose(1, 2)
    l1_xyz, l1 = _sa(model.sa1, xyz, pts)
    l2_xyz, l2 = _sa(model.sa2, l1_xyz, l1)
    l3_xyz, l3 = _sa(model.sa3, l2_xyz, l2)
    l2 = _fp(model.fp3, l2_xyz, l3_xyz, l2, l3)
    l1 = _fp(model.fp2, l1_xyz, l2_xyz, l1, l2)
    l0 = _fp(model.fp1, xyz, l1_xyz, pts if model.rgb_skip else None, l1)
    return model.conv2(model.drop1(F.relu(model.bn1(model.conv1(l0)))))


def pointnet2_msg(model, xyz, colors):
    """SA/FP trunk of EnhancedPointNet2 (models/model.py:128-147) on the product's container."""
    pts = colors.transpose(1, 2)
    l1_xyz, l1 = _sa(model.sa1, xyz, pts)
    l2_xyz, l2 = _sa(model.sa2, l1_xyz, l1)
    l3_xyz, l3 = _sa(model.sa3, l2_xyz, l2)
    l2 = _fp(model.fp3, l2_xyz, l3_xyz, l2, l3)
    l1 = _fp(model.fp2, l1_xyz, l2_xyz, l1, l2)
    l0 = _fp(model.fp1, xyz, l1_xyz, pts, l1)
    n = l0.shape[2]
    # MultiScaleFeatureFusion.forward, models/model.py:160-167: nearest resampling to N, conv each, cat
    fused = torch.cat([conv(F.interpolate(f, size=n)) for f, conv in zip([l2, l1, l0], model.fusion.convs)], dim=1)
    return model.final_fusion(fused)


def graph_feature(x_bdn, k):
    """DGCNN.get_graph_feature, DGCNN.py:72-109."""
    B, D, N = x_bdn.shape
    idx = knn_graph(x_bdn, k)
    xt = x_bdn.transpose(2, 1).contiguous()
    nb = take_rows(xt, idx)
    ctr = xt.view(B, N, 1, D).expand(-1, -1, k, -1)
    return torch.cat((nb - ctr, ctr), dim=3).permute(0, 3, 1, 2).contiguous()


def dgcnn(model, xyz, colors=None):
    """DGCNN.forward, DGCNN.py:111-172."""
    N = xyz.shape[1]
    x = xyz.transpose(2, 1)[:, :3, :]
    k = min(model.k, N - 1)
    feats = []
    for block in (model.conv1, model.conv2, model.conv3, model.conv4):
        x = block(graph_feature(x, k)).max(dim=-1)[0]
        feats.append(x)
    local = torch.cat(feats, dim=1)
    local_n = F.leaky_relu(model.local_bn(local), negative_slope=0.2)
    g = F.adaptive_max_pool1d(model.conv5(local), 1)
    return model.point_conv(torch.cat([local_n, g.expand(-1, -1, N)], dim=1)).transpose(1, 2)


# ----------------------------------------------------------------------------- bridge encoders
def neighbourhood_descriptor(rel):
    """BridgeStructureEncoding.get_structure_features, attention_modules.py:620-687, with the same
    ATen operations: rel [B,N,k,3] -> [B,N,13]."""
    B, N, k, _ = rel.shape
    flat = rel.reshape(B * N, k, 3)
    second = torch.bmm(flat.transpose(1, 2), flat) / (k - 1)                       # :631
    ev = torch.linalg.eigh(second)[0].view(B, N, 3)                                # ascending
    shape3 = torch.stack([(ev[..., 0] - ev[..., 1]) / (ev[..., 0] + 1e-8),
                          (ev[..., 1] - ev[..., 2]) / (ev[..., 0] + 1e-8),
                          ev[..., 2] / (ev[..., 0] + 1e-8)], dim=-1)               # :637-641
    spread = torch.norm(rel - rel.mean(dim=2, keepdim=True), dim=-1)               # :646-647
    stats3 = torch.stack([spread.max(dim=-1)[0], spread.mean(dim=-1), spread.std(dim=-1)], dim=-1)
    unit = (rel / (torch.norm(rel, dim=-1, keepdim=True) + 1e-8)).reshape(B * N, k, 3)
    cosine = torch.bmm(unit, unit.transpose(1, 2)).view(B, N, k, k).mean(dim=(-1, -2))  # :657-662
    z = rel[..., 2]
    z2 = torch.stack([z.std(dim=-1), z.max(dim=-1)[0] - z.min(dim=-1)[0]], dim=-1)  # :665-668
    return torch.cat([shape3, stats3, cosine.unsqueeze(-1), z2, rel.mean(dim=2),
                      torch.norm(rel.std(dim=2), dim=-1, keepdim=True)], dim=-1)   # :674-681


def cdist_neighbours(xyz, k):
    """torch.cdist + topk(largest=False), attention_modules.py:584-586 -> [B,N,k] int64."""
    return torch.cdist(xyz, xyz).topk(k, dim=-1, largest=False)[1]


def structure_encoding(mod, xyz):
    """BridgeStructureEncoding.forward, attention_modules.py:577-618 -> [B,channels,N]."""
    B, N, _ = xyz.shape
    k = min(mod.k, N)
    grid = torch.floor(xyz / mod.grid_size) * mod.grid_size
    absolute = torch.cat([f(grid * w) for w in mod.freqs for f in (torch.sin, torch.cos)], dim=-1)
    idx = cdist_neighbours(xyz, k)
    rel = take_rows(xyz, idx) - xyz.unsqueeze(2)
    desc = neighbourhood_descriptor(rel)
    x = torch.cat([absolute.unsqueeze(2).expand(-1, -1, k, -1), rel,
                   desc.unsqueeze(2).expand(-1, -1, k, -1)], dim=-1).permute(0, 3, 1, 2)
    return mod.structure_mlp(x).max(dim=-1)[0]


def geometric_extraction(mod, x, xyz):
    """GeometricFeatureExtraction.forward, attention_modules.py:253-269."""
    return mod.mlp(torch.cat([x, structure_encoding(mod.br_pos, xyz)], dim=1))


def colour_extraction(mod, colors, xyz):
    """ColorFeatureExtraction.forward, attention_modules.py:718-753 (its cdist + topk + gather,
    :736-743, are executed for their cost and, as there, not used)."""
    feat = mod.color_mlp(colors)
    B, _, N = feat.shape
    idx = cdist_neighbours(xyz, 16)
    take_rows(feat.transpose(1, 2), idx)
    local = feat * mod.color_attention(feat)
    return local * mod.color_context(feat)


def bridgeseg(model, xyz, colors):
    """EnhancedPointNet2.forward, models/model.py:113-147, on the product's container."""
    pos = structure_encoding(model.bri_enc, xyz)
    col = colour_extraction(model.color_encoder, colors.transpose(1, 2), xyz)
    pts = model.feature_fusion.fusion_mlp(torch.cat([pos, col], dim=1))
    l1_xyz, l1 = _sa(model.sa1, xyz, pts)
    l2_xyz, l2 = _sa(model.sa2, l1_xyz, l1)
    l2 = geometric_extraction(model.geometric2, l2, l2_xyz)
    l3_xyz, l3 = _sa(model.sa3, l2_xyz, l2)
    l3 = geometric_extraction(model.geometric3, l3, l3_xyz)
    l2 = _fp(model.fp3, l2_xyz, l3_xyz, l2, l3)
    l1 = _fp(model.fp2, l1_xyz, l2_xyz, l1, l2)
    l0 = _fp(model.fp1, xyz, l1_xyz, pts, l1)
    n = l0.shape[2]
    fused = torch.cat([conv(F.interpolate(f, size=n)) for f, conv in zip([l2, l1, l0], model.fusion.convs)], dim=1)
    return model.final_fusion(fused)


def run(model, xyz, colors):
    """Forward the product's container `model` (on CPU) through the ATen port."""
    name = type(model).__name__
    if name == "PointNet2":
        return pointnet2(model, xyz, colors)
    if name == "PointNet2MSG":
        return pointnet2_msg(model, xyz, colors)
    if name == "DGCNN":
        return dgcnn(model, xyz, colors)
    if name == "EnhancedPointNet2":
        return bridgeseg(model, xyz, colors)
    raise TypeError(f"no CPU port for {name}")
