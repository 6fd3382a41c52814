"""Debug: which side is wrong in the fp32 two-layer SyncBatchNorm case -- single process or sharded?
Run under torch.distributed.run with 2 ranks (gloo)."""
import copy
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloud_bridge_amd import parallel, rowmlp  # noqa: E402

rank, world, _ = parallel.init_from_env("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rowmlp.set_precision("fp32")
torch.manual_seed(3)
R, K, widths = 2048, 264, [256, 128]
convs = nn.ModuleList(nn.Conv2d(a, b, 1) for a, b in zip([K] + widths[:-1], widths)).to(dev)
bns = nn.ModuleList(nn.BatchNorm2d(b) for b in widths).to(dev).train()
with torch.no_grad():
    for bn in bns:
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
x_all = torch.randn(world * R, K, device=dev)
g_all = torch.randn(world * R, widths[-1], device=dev)
sync_bns = nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(bns))
sync_convs = copy.deepcopy(convs)
x = x_all[rank * R:(rank + 1) * R].clone().requires_grad_(True)
out = rowmlp.mlp_rows(sync_convs, sync_bns, x, pool=0)
(out * g_all[rank * R:(rank + 1) * R]).sum().backward()
grads = [p.grad.clone() for mod in (sync_convs, sync_bns) for p in mod.parameters()]
for t in grads:
    dist.all_reduce(t)
names = [n for mod, tag in ((sync_convs, "conv"), (sync_bns, "bn")) for n, _ in mod.named_parameters(prefix=tag)]
if rank == 0:
    xf = x_all.clone().requires_grad_(True)
    ref = rowmlp.mlp_rows(convs, bns, xf, pool=0)
    (ref * g_all).sum().backward()
    single = [p.grad.clone() for mod in (convs, bns) for p in mod.parameters()]
    # fp64 truth
    x64 = x_all.double().requires_grad_(True)
    h = x64
    c64 = copy.deepcopy(convs).double()
    b64 = nn.ModuleList(nn.BatchNorm1d(w) for w in widths).to(dev).double().train()
    for bb, bo in zip(b64, bns):
        bb.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bo.state_dict().items()})
        bb.reset_running_stats()
    for c, b in zip(c64, b64):
        h = F.relu(b(F.linear(h, c.weight.view(c.out_channels, -1), c.bias)))
    (h * g_all.double()).sum().backward()
    truth = [p.grad for mod in (c64, b64) for p in mod.parameters()]

    def err(a, b):
        return float((a.double().reshape(-1) - b.double().reshape(-1)).abs().max() / b.double().abs().max().clamp_min(1e-9))
    for n, s, g, t in zip(names, single, grads, truth):
        print(f"{n:16s} single vs fp64 {err(s, t):.2e}   sharded vs fp64 {err(g, t):.2e}")
    print(f"dx               single vs fp64 {err(xf.grad, x64.grad):.2e}   sharded(rank0 rows) vs fp64 {err(x.grad, x64.grad[:R]):.2e}")
dist.barrier()
dist.destroy_process_group()
