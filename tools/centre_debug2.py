import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloud_bridge_amd import _lib
from pointcloud_bridge_amd.ops import _launch
L = _lib.load()
torch.manual_seed(0)
for (R, N, K, pro) in ((900, 512, 384, 0), (900, 64, 8, 0), (3000, 256, 72, 0), (900, 320, 512, 1), (1000, 128, 640, 1), (5000, 264, 264, 0)):
    a = torch.randn(R, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.2).to(torch.bfloat16)
    scale = torch.rand(K, device="cuda") + 0.5
    shift = torch.randn(K, device="cuda") * 0.3
    centre = torch.randn(N, device="cuda") * 2
    out = torch.empty(R, N, dtype=torch.bfloat16, device="cuda")
    nparts = 13
    sums = torch.full((nparts, 2, N), float("nan"), device="cuda")
    _launch("pcb_gemm_nt_stats_bf16", 0, pro, a.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, w.data_ptr(), R, N, K,
            out.data_ptr(), sums.data_ptr(), nparts, centre.data_ptr())
    x = a.float()
    if pro:
        x = torch.relu(x * scale + shift).to(torch.bfloat16).float()
    ref = x @ w.float().t() - centre
    d = (out.float() - ref).abs()
    tot = sums.double().sum(0)
    e0 = (tot[0] - out.double().sum(0)).abs() / (out.double().abs().sum(0) + 1e-9)
    e1 = (tot[1] - (out.double() ** 2).sum(0)).abs() / ((out.double() ** 2).sum(0) + 1e-9)
    print(R, N, K, pro, "max err", float(d.max()), "of", float(ref.abs().max()), "stats rel err", float(e0.max()), float(e1.max()), "nan", bool(torch.isnan(tot).any()))
