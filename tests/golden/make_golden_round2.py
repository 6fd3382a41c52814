"""Round-2 golden vectors from the REFERENCE (CPU, fp32), run in the build container only:

  sample_and_group.npz   sample_and_group(npoint, radius, nsample, xyz, points)
                         (models/pointnet2_utils.py:42-60) on a continuous and on a grid cloud:
                         new_xyz, new_points, with the CPU-generator seed in front of the call.
  train_steps.npz        SURVEY section 8 row H1, "loss after a few optimiser steps": the reference's
                         PointNet2 (models/model.py:12-56) under the reference loop's optimiser --
                         Adam(lr=1e-3, betas=(.9,.999), weight_decay=1e-4), train_MulSca_PN2.py:125 --
                         and criterion (CrossEntropyLoss on [B,C,N], :161) for 4 steps over two fixed
                         batches: the loss of every step, the eval-mode logits after the last one.
                         Dropout is kept in eval mode (no RNG but FPS's torch.randint is consumed).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_round2.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Highway_bridge"
sys.dont_write_bytecode = True
for p in (REF, REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from make_golden import unit_ball_cloud  # noqa: E402

INIT_SEED = 42
FWD_SEED = 321


def make_sample_and_group():
    from models import pointnet2_utils as rpu
    g = torch.Generator().manual_seed(5)
    B, N, C = 2, 640, 5
    cont = unit_ball_cloud(g, B, N)
    grid = (torch.randint(-1024, 1024, (B, N, 3), generator=g).float() / 1024.0).contiguous()
    pts = torch.randn(B, N, C, generator=g)
    out = {"points": pts.numpy(), "fwd_seed": np.int64(FWD_SEED), "npoint": np.int64(96),
           "radius": np.float64(0.3), "nsample": np.int64(12)}
    for tag, xyz in (("cont", cont), ("grid", grid)):
        torch.manual_seed(FWD_SEED)
        new_xyz, new_points = rpu.sample_and_group(96, 0.3, 12, xyz, pts)
        torch.manual_seed(FWD_SEED)
        nx0, np0 = rpu.sample_and_group(96, 0.3, 12, xyz, None)
        out.update({f"{tag}_xyz": xyz.numpy(), f"{tag}_new_xyz": new_xyz.numpy(), f"{tag}_new_points": new_points.numpy(),
                    f"{tag}_new_points_nofeat": np0.numpy()})
        assert torch.equal(nx0, new_xyz)
    np.savez_compressed(os.path.join(HERE, "sample_and_group.npz"), **out)
    print("sample_and_group.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.shape})


def make_train_steps():
    from models.model import PointNet2 as RefSSG
    from pointcloud_bridge_amd.models.containers import PointNet2
    g = torch.Generator().manual_seed(23)
    B, N, steps = 2, 1024, 4
    batches = []
    for _ in range(2):
        xyz = unit_ball_cloud(g, B, N)
        colors = torch.rand(B, N, 3, generator=g)
        labels = ((xyz[:, :, 2] + 1.0) * 2.5).long().clamp_(0, 4)  # a learnable task: horizontal slabs
        batches.append((xyz, colors, labels))
    torch.manual_seed(INIT_SEED)
    ref = RefSSG(5)
    torch.manual_seed(INIT_SEED)
    mine = PointNet2(5)
    a, b = ref.state_dict(), mine.state_dict()
    assert list(a.keys()) == list(b.keys()) and all(torch.equal(a[k], b[k]) for k in a)
    for s in ref.modules():
        if isinstance(s, nn.Dropout):
            s.eval()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)  # train_MulSca_PN2.py:125
    losses = []
    ref.train()
    for s in ref.modules():
        if isinstance(s, nn.Dropout):
            s.eval()
    torch.manual_seed(FWD_SEED)
    for i in range(steps):
        xyz, colors, labels = batches[i % 2]
        opt.zero_grad()
        loss = F.cross_entropy(ref(xyz, colors), labels)  # :161
        loss.backward()
        opt.step()
        losses.append(loss.item())
    ref.eval()
    with torch.no_grad():
        final = ref(batches[0][0], batches[0][1]).numpy()
    out = {"init_seed": np.int64(INIT_SEED), "fwd_seed": np.int64(FWD_SEED), "steps": np.int64(steps),
           "losses": np.array(losses, dtype=np.float64), "final_logits_eval": final}
    for i, (xyz, colors, labels) in enumerate(batches):
        out.update({f"xyz{i}": xyz.numpy(), f"colors{i}": colors.numpy(), f"labels{i}": labels.numpy()})
    np.savez_compressed(os.path.join(HERE, "train_steps.npz"), **out)
    print("train_steps.npz losses", losses)


if __name__ == "__main__":
    torch.set_num_threads(8)
    make_sample_and_group()
    make_train_steps()
