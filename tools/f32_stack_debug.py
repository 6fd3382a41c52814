"""Debug the fp32 stack at R=8192, K=64, widths [256,128]: check the prepared operands and the
per-layer constants against fp64 torch."""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloud_bridge_amd import rowmlp  # noqa: E402

R, K, widths = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 64, [256, 128]
dev = "cuda"
torch.manual_seed(R + K)
rowmlp.set_precision("fp32")
x = torch.randn(R, K, device=dev).requires_grad_(True)
convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip([K] + widths[:-1], widths)).to(dev)
bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).to(dev).train()
with torch.no_grad():
    for bn in bns:
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
out = rowmlp.mlp_rows(convs, bns, x, 1, 0, 0)
xs, arg, ybuf, stz, wbuf = out.grad_fn.saved_tensors[:5]
g = torch.randn_like(out)
(out * g).sum().backward()
torch.cuda.synchronize()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))


C0, C1 = widths
w0 = convs[0].weight.view(C0, K)
w1 = convs[1].weight.view(C1, C0)
off = 0
wp0 = wbuf[off:off + C0 * K].view(C0, K); off += C0 * K
wt0 = wbuf[off:off + C0 * K].view(K, C0); off += C0 * K
wp1 = wbuf[off:off + C1 * C0].view(C1, C0); off += C1 * C0
wt1 = wbuf[off:off + C1 * C0].view(C0, C1); off += C1 * C0
print("wp0", rel(wp0, w0), "wt0", rel(wt0, w0.t()), "wp1", rel(wp1, w1), "wt1", rel(wt1, w1.t()), "wbuf numel", wbuf.numel(), off)
y0 = ybuf[:R * C0].view(R, C0)
y1 = ybuf[R * C0:R * C0 + R * C1].view(R, C1)
x64 = x.detach().double()
y0r = x64 @ w0.double().t()
print("y0", rel(y0, y0r))
st0 = stz[:10 * C0].view(10, C0)
st1 = stz[10 * C0:10 * C0 + 10 * C1].view(10, C1)
m0, v0 = y0r.mean(0), y0r.var(0, unbiased=False)
sc0 = bns[0].weight.double() / torch.sqrt(v0 + 1e-5)
sh0 = bns[0].bias.double() - m0 * sc0
print("L0 scale", rel(st0[2], sc0), "shift", rel(st0[3], sh0), "mean", rel(st0[4], m0))
z0 = F.relu(y0r * sc0 + sh0)
y1r = z0 @ w1.double().t()
print("y1", rel(y1, y1r))
m1, v1 = y1r.mean(0), y1r.var(0, unbiased=False)
is1 = 1 / torch.sqrt(v1 + 1e-5)
sc1 = bns[1].weight.double() * is1
sh1 = bns[1].bias.double() - m1 * sc1
print("L1 scale", rel(st1[2], sc1), "shift", rel(st1[3], sh1))
du1 = g.double() * ((y1r * sc1 + sh1) > 0)
s1, s2 = du1.sum(0), (du1 * (y1r - m1) * is1).sum(0)
p1 = -sc1 * is1 * s2 / R
q1 = -sc1 * s1 / R - p1 * m1
print("L1 p", rel(st1[8], p1), "q", rel(st1[9], q1))
dy1 = sc1 * du1 + p1 * y1r + q1
dz0 = dy1 @ w1.double()
is0 = 1 / torch.sqrt(v0 + 1e-5)
du0 = dz0 * ((y0r * sc0 + sh0) > 0)
t1, t2 = du0.sum(0), (du0 * (y0r - m0) * is0).sum(0)
p0 = -sc0 * is0 * t2 / R
q0 = -sc0 * t1 / R - p0 * m0
print("L0 p", rel(st0[8], p0), "q", rel(st0[9], q0), "dbeta0", rel(bns[0].bias.grad, t1), "dgamma0", rel(bns[0].weight.grad, t2))
dy0 = sc0 * du0 + p0 * y0r + q0
print("dW0", rel(convs[0].weight.grad.view(C0, K), dy0.t() @ x64), "dx", rel(x.grad, dy0 @ w0.double()))
# where is dbeta0 wrong?  per-channel relative error
e = ((bns[0].bias.grad.double() - t1).abs() / t1.abs().max())
print("dbeta0 worst channels", torch.topk(e, 8).indices.tolist(), [f"{v:.1e}" for v in torch.topk(e, 8).values.tolist()])
