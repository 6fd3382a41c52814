"""Direct checks of single fp32 entry points against fp64 torch at the sizes where a stack went wrong."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloud_bridge_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


for R in (4096, 8192, 8192 + 64):
    for (N, K) in ((256, 128), (64, 256), (264, 256)):
        dz = torch.randn(R, K, device=dev)
        y = torch.randn(R, K, device=dev)
        scale, shift = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
        p, q = torch.randn(K, device=dev) * 0.1, torch.randn(K, device=dev) * 0.1
        w = torch.randn(N, K, device=dev)
        out = torch.full((R, N), float("nan"), device=dev)
        rc = lib.pcb_gemm_nt_f32(2, dz.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), p.data_ptr(), q.data_ptr(), 0, 0,
                                 1, 1, w.data_ptr(), R, N, K, out.data_ptr(), 0, 0, st)
        torch.cuda.synchronize()
        dy = scale.double() * dz.double() * ((y.double() * scale.double() + shift.double()) > 0) + p.double() * y.double() + q.double()
        print(f"gemm_nt_f32 pro=2 R={R} N={N} K={K}: rc {rc} err {rel(out, dy @ w.double().t()):.2e}")
    for C in (128, 256):
        dz = torch.randn(R, C, device=dev)
        y = torch.randn(R, C, device=dev)
        scale, shift = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
        mean, invstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        sums = torch.zeros(2, C, device=dev)
        rc = lib.pcb_bn_act_bwd_reduce_f32(dz.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                           invstd.data_ptr(), R, C, 1, sums.data_ptr(), 1, st)
        torch.cuda.synchronize()
        du = dz.double() * ((y.double() * scale.double() + shift.double()) > 0)
        s1, s2 = du.sum(0), (du * (y.double() - mean.double()) * invstd.double()).sum(0)
        print(f"bwd_reduce_f32 R={R} C={C}: rc {rc} err s1 {rel(sums[0], s1):.2e} s2 {rel(sums[1], s2):.2e}")
