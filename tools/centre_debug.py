"""Debug: one fused stack in bf16 rows with / without centred storage against an fp32 torch evaluation."""
import os, sys
import torch, torch.nn as nn, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloud_bridge_amd import rowmlp as rm

def run(R, K, widths, centring, calls=1):
    torch.manual_seed(R + K)
    dev = "cuda"
    x = torch.randn(R, K, device=dev).to(torch.bfloat16)
    dims = [K] + widths
    convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip(dims[:-1], dims[1:])).to(dev)
    bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).to(dev).train()
    old = rm.set_centring(centring)
    with rm.precision("bf16"):
        for _ in range(calls):
            out = rm.mlp_rows(convs, bns, x, 1, 0, 0)
    rm.set_centring(old)
    h = x.float()
    for conv, bn in zip(convs, bns):
        w = conv.weight.detach().view(conv.out_channels, -1).to(torch.bfloat16).float()
        z = h @ w.t()
        zn = (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + bn.eps)
        h = F.relu(zn * bn.weight + bn.bias).to(torch.bfloat16).float()
    d = (out.float() - h).abs()
    cols = d.max(0)[0]
    print(f"R={R} K={K} widths={widths} centring={centring} calls={calls}: max err {float(d.max()):.4f} of {float(h.abs().max()):.3f}; "
          f"worst columns {cols.topk(5)[1].tolist()} col-err by 64-block {[round(float(cols[i:i+64].max()),3) for i in range(0, cols.numel(), 64)]}")
    for i, bn in enumerate(bns):
        c = getattr(bn, "_pcb_centre", None)
        if c is not None:
            print("   layer", i, "centre abs max", float(c.abs().max()), "running_mean abs max", float(bn.running_mean.abs().max()))

for cfg in ((900, 384, [512, 320]), (700, 320, [384, 64]), (1100, 264, [256, 256]), (3000, 72, [256, 128]), (2048, 8, [64, 64, 128])):
    for centring in (False, True):
        run(*cfg, centring)
    run(*cfg, True, calls=2)
