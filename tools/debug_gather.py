import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import rowmlp as rm
from pointcloud_bridge_amd.models import pointnet2_utils as pu
dev = torch.device("cuda")
torch.manual_seed(9)
B, N = 2, 512
xyz = torch.rand(B, N, 3, device=dev) * 2 - 1
feat = torch.randn(B, 64, N, device=dev).to(torch.bfloat16).float()
rm.set_precision("bf16")
torch.manual_seed(0)
mod = pu.SetAbstraction(64, 0.4, 16, 64 + 3, [32, 32, 64]).to(dev).train()
def run(g, mode="bf16"):
    rm.set_precision(mode); rm.set_gathered(g); mod.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    leaf = feat.clone().requires_grad_(True)
    out = mod(xyz, leaf)[1]
    w = torch.linspace(-1.0, 1.0, out.numel(), device=dev).view_as(out)
    (out.float() * w).sum().backward()
    return out.detach().float(), leaf.grad.float(), mod.mlp_convs[0].weight.grad.clone().view(32, 67), mod.mlp_convs[1].weight.grad.clone().view(32, 32)
a, b, c = run(True), run(False), run(False, "fp32")
for name, x, y in (("gath vs group", a, b), ("gath vs fp32", a, c), ("group vs fp32", b, c)):
    print(name, "out", float((x[0]-y[0]).abs().mean()/y[0].abs().mean()), "dfeat", float((x[1]-y[1]).abs().mean()/y[1].abs().mean()),
          "dW0 xyz", float((x[2][:, :3]-y[2][:, :3]).abs().mean()/y[2][:, :3].abs().mean()),
          "dW0 feat", float((x[2][:, 3:]-y[2][:, 3:]).abs().mean()/y[2][:, 3:].abs().mean()),
          "dW1", float((x[3]-y[3]).abs().mean()/y[3].abs().mean()))
