"""Torch-only check (no libpcb_hip.so involved): a captured ATen reduction over the middle dimension
of a [2, 768, 1024] tensor, replayed (a) back to back and (b) with a device synchronisation and
ordinary allocations in between.  DGCNN's global pool and the backward of its expand+cat are such
reductions; the captured DGCNN step went wrong exactly there on replays of kind (b)."""
import torch

dev = "cuda"
torch.manual_seed(0)
base = torch.randn(2, 768, 1344, device=dev)
x = base[:, :, 320:]                       # the strided slice cat-backward hands to expand-backward
ref_sum = x.sum(dim=1, keepdim=True).clone()
ref_max, ref_idx = x.max(dim=1, keepdim=True)
out_sum = torch.empty_like(ref_sum)
out_max = torch.empty_like(ref_max)
out_idx = torch.empty_like(ref_idx)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    out_sum.copy_(x.sum(dim=1, keepdim=True))
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out_sum.copy_(x.sum(dim=1, keepdim=True))
    m, i = x.max(dim=1, keepdim=True)
    out_max.copy_(m)
    out_idx.copy_(i)


def check(tag):
    torch.cuda.synchronize()
    es = float((out_sum - ref_sum).abs().max())
    em = float((out_max - ref_max).abs().max())
    ei = int((out_idx != ref_idx).sum())
    lo, hi = int(out_idx.min()), int(out_idx.max())
    print(f"{tag}: sum err {es:.3e}  max err {em:.3e}  index mismatches {ei}  index range [{lo}, {hi}]")


for r in range(3):
    base.normal_()                          # new input every replay: a stale output cannot pass
    ref_sum = x.sum(dim=1, keepdim=True).clone()
    ref_max, ref_idx = x.max(dim=1, keepdim=True)
    g.replay()
    check(f"replay {r} (after eager work + sync)")
    junk = [torch.randn(257, 1031, device=dev) for _ in range(8)]
    del junk
