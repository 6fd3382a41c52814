import sys, os, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import rowmlp as rm
def ste(y): return y + (y.to(torch.bfloat16).float() - y).detach()
def act_(u, a): return F.relu(u) if a == 1 else (F.leaky_relu(u, 0.2) if a == 2 else u)
def rel(a, b):
    d = (a.float() - b).abs(); return float(d.mean() / b.abs().mean().clamp_min(1e-9)), float(d.max() / b.abs().max().clamp_min(1e-9))
for (R, K, widths, act, pool) in [(2048, 8, [64, 64], 1, 0), (2048, 8, [64, 64], 1, 16), (2048, 8, [64, 64, 128], 1, 16), (3000, 72, [256, 128], 1, 0)]:
    torch.manual_seed(0); dev = "cuda"
    x = torch.randn(R, K, device=dev).to(torch.bfloat16).requires_grad_(True)
    dims = [K] + widths
    convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip(dims[:-1], dims[1:])).to(dev)
    bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).to(dev).train()
    refs = nn.ModuleList(nn.BatchNorm1d(b) for b in widths).to(dev).train()
    for bn, rf in zip(bns, refs): rf.load_state_dict(bn.state_dict())
    rm.set_precision("bf16"); out = rm.mlp_rows(convs, bns, x, act, pool, 0); rm.set_precision("fp32")
    xr = x.detach().float().requires_grad_(True); h = xr; ws = []
    for i, (conv, rf) in enumerate(zip(convs, refs)):
        w = conv.weight.detach().view(conv.out_channels, -1).to(torch.bfloat16).float().requires_grad_(True); ws.append(w)
        h = act_(rf(ste(h @ w.t()) + conv.bias.detach()), act)
        if i < len(convs) - 1: h = ste(h)
    ref = h.view(-1, pool, widths[-1]).max(dim=1)[0] if pool else h
    g = torch.randn_like(ref)
    (out.float() * g).sum().backward(); (ref * g).sum().backward()
    print(f"case R={R} K={K} widths={widths} pool={pool}: out", rel(out, ref.detach()), "dx", rel(x.grad, xr.grad))
    for i, (conv, w, bn, rf) in enumerate(zip(convs, ws, bns, refs)):
        print(f"   layer {i}: dW", rel(conv.weight.grad.view_as(w), w.grad), "dgamma", rel(bn.weight.grad, rf.weight.grad), "dbeta", rel(bn.bias.grad, rf.bias.grad))

# grouped-layout case (features first, then xyz, zero padding) ---------------------------------
R, K, widths, act, pool, perm = 1536, 24, [32, 32, 64], 1, 8, 16
torch.manual_seed(R + K); dev = "cuda"; kin = perm + 3
x = torch.randn(R, K, device=dev).to(torch.bfloat16); x[:, kin:] = 0; x.requires_grad_(True)
dims = [kin] + widths
convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip(dims[:-1], dims[1:])).to(dev)
bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).to(dev).train()
refs = nn.ModuleList(nn.BatchNorm1d(b) for b in widths).to(dev).train()
for bn, rf in zip(bns, refs): rf.load_state_dict(bn.state_dict())
rm.set_precision("bf16"); out = rm.mlp_rows(convs, bns, x, act, pool, perm); rm.set_precision("fp32")
xr = x.detach().float().requires_grad_(True); h = xr[:, :kin]; h = torch.cat([h[:, perm:perm + 3], h[:, :perm]], dim=1); ws = []
for i, (conv, rf) in enumerate(zip(convs, refs)):
    w = conv.weight.detach().view(conv.out_channels, -1).to(torch.bfloat16).float().requires_grad_(True); ws.append(w)
    h = act_(rf(ste(h @ w.t()) + conv.bias.detach()), act)
    if i < len(convs) - 1: h = ste(h)
ref = h.view(-1, pool, widths[-1]).max(dim=1)[0]
g = torch.randn_like(ref)
(out.float() * g).sum().backward(); (ref * g).sum().backward()
print("perm case: out", rel(out, ref.detach()), "dx", rel(x.grad[:, :kin], xr.grad[:, :kin]), "dx pad", float(x.grad[:, kin:].abs().max()))
for i, (conv, w, bn, rf) in enumerate(zip(convs, ws, bns, refs)):
    print(f"   layer {i}: dW", rel(conv.weight.grad.view_as(w), w.grad), "dgamma", rel(bn.weight.grad, rf.weight.grad))
d = (x.grad[:, :kin].float() - xr.grad[:, :kin]).abs()
print("worst rows", d.max(dim=1)[0].topk(5), "worst cols", d.max(dim=0)[0])
