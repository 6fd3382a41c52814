"""Golden vectors of scope row f4 from the REFERENCE (CPU, fp32), run in the build container only:

  model_ptv3.npz   PointTransformerV3 (models/PointTransformerV3.py:173-305) in the configuration of
                   inference_ptv3.py:101-105 (embed 384, 2 heads -> head_dim 192, qkv_bias, mlp_ratio 4) with
                   depth 3 (the block is the same at every depth; 8 blocks would only make the fixture slower),
                   eval mode, on B=2 clouds of N=640 points with colours: logits [B, N, 5].
                   The drop-in is constructed under the same seed and its state_dict compared key by key.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ptv3.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Highway_bridge"
sys.dont_write_bytecode = True
for p in (REF, REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from make_golden import unit_ball_cloud  # noqa: E402

INIT_SEED = 77
CFG = dict(num_classes=5, d_in=6, embed_dim=384, depth=3, num_heads=2, mlp_ratio=4., qkv_bias=True, drop_rate=0.1,
           attn_drop_rate=0.1)


def main():
    from models.PointTransformerV3 import PointTransformerV3 as Ref
    from pointcloud_bridge_amd.models.PointTransformerV3 import PointTransformerV3 as Mine
    torch.manual_seed(INIT_SEED)
    ref = Ref(**CFG).eval()
    torch.manual_seed(INIT_SEED)
    mine = Mine(**CFG).eval()
    sa, sb = ref.state_dict(), mine.state_dict()
    assert list(sa) == list(sb), "state_dict keys differ"
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    g = torch.Generator().manual_seed(9)
    xyz = unit_ball_cloud(g, 2, 640)
    colors = torch.rand(2, 640, 3, generator=g)
    # non-trivial BatchNorm running statistics in the head (fresh ones would make it an identity up to eps)
    with torch.no_grad():
        ref.head[1].running_mean.uniform_(-0.2, 0.2, generator=g)
        ref.head[1].running_var.uniform_(0.5, 1.5, generator=g)
        logits = ref(xyz, colors)
        logits_xyz_only = Ref(**dict(CFG, d_in=3)).eval()  # shape check of the d_in = 3 variant only
        assert logits_xyz_only(xyz).shape == (2, 640, 5)
    out = {"xyz": xyz.numpy(), "colors": colors.numpy(), "logits_eval": logits.numpy(), "init_seed": np.int64(INIT_SEED),
           "head_running_mean": ref.head[1].running_mean.numpy(), "head_running_var": ref.head[1].running_var.numpy(),
           "depth": np.int64(CFG["depth"]), "num_state_keys": np.int64(len(sa))}
    np.savez_compressed(os.path.join(HERE, "model_ptv3.npz"), **out)
    print("model_ptv3.npz", {k: getattr(v, "shape", None) for k, v in out.items()}, "keys", len(sa),
          "logit scale", float(logits.abs().max()))


if __name__ == "__main__":
    main()
