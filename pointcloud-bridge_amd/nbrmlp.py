"""Neighbourhood MLP of BridgeStructureEncoding on the HIP kernels of csrc/nbrmlp.hip.

Reference: models/attention_modules.py:548-553, 606-616 -- Conv2d 1x1 -> BatchNorm2d -> ReLU ->
Conv2d 1x1 over the expanded [B, 40, N, k] tensor, then the max over k.  Here the rows are never
materialised: `base` [P, C] carries the per-point part of the first convolution, the kernels add the
3 offset channels per neighbour, apply the folded BatchNorm + ReLU + second convolution and keep the
running maximum in registers.  fp32; forward = statistics pass + finalize + apply pass, backward =
reduce pass + finalize + apply pass (the same BatchNorm-backward algebra as the fused bf16 stacks).
"""
import torch

from . import _lib
from .ops import _launch, on_device, sum_slabs
from .rowmlp import _bn_bookkeeping, _counter


class _NeighbourMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, rel, wr, gamma, beta, w2, b2, running_mean, running_var, training, momentum, eps,
                counter):
        P, C = base.shape
        k = rel.shape[1]
        dev = base.device
        base, rel = base.contiguous(), rel.contiguous()
        wr, w2 = wr.contiguous(), w2.contiguous()
        parts = _lib.load().pcb_nbr_mlp_partials(P)
        consts = torch.empty(4, C, dtype=torch.float32, device=dev)  # scale | shift | mean | invstd
        out = torch.empty(P, C, dtype=torch.float32, device=dev)
        arg = torch.empty(P, C, dtype=torch.uint8, device=dev)
        with on_device(dev):
            sums = None
            if training:
                sums = torch.empty(parts, 2, C, dtype=torch.float32, device=dev)
                _launch("pcb_nbr_mlp_stats", P * k * C, base.data_ptr(), rel.data_ptr(), P, k, C, wr.data_ptr(),
                        sums.data_ptr())
            _launch("pcb_bn_finalize", C, 0 if sums is None else sums.data_ptr(), parts, P * k, 0, C,
                    0 if gamma is None else gamma.data_ptr(), 0 if beta is None else beta.data_ptr(), 0,
                    0 if running_mean is None else running_mean.data_ptr(),
                    0 if running_var is None else running_var.data_ptr(),
                    float(momentum), float(eps), int(training), consts[0].data_ptr(), consts[1].data_ptr(),
                    consts[2].data_ptr(), consts[3].data_ptr(), 0 if counter is None else counter.data_ptr())
            _launch("pcb_nbr_mlp_forward", P * k * C * C, base.data_ptr(), rel.data_ptr(), P, k, C, wr.data_ptr(),
                    consts[0].data_ptr(), consts[1].data_ptr(), w2.data_ptr(), 0 if b2 is None else b2.data_ptr(),
                    out.data_ptr(), arg.data_ptr())
        ctx.save_for_backward(base, rel, wr, w2, consts, arg)
        ctx.cfg = (int(training), parts, gamma is not None, b2 is not None)
        ctx.mark_non_differentiable(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        base, rel, wr, w2, consts, arg = ctx.saved_tensors
        training, parts, has_affine, has_b2 = ctx.cfg
        P, C = base.shape
        k = rel.shape[1]
        dev = base.device
        g = g.contiguous().float()
        sums = torch.empty(parts, 2, C, dtype=torch.float32, device=dev)
        dw2p = torch.empty(parts, C, C + 1, dtype=torch.float32, device=dev)
        dwrp = torch.empty(parts, C, 3, dtype=torch.float32, device=dev)
        pq = torch.empty(4, C, dtype=torch.float32, device=dev)  # p | q | dgamma | dbeta
        dbase = torch.empty(P, C, dtype=torch.float32, device=dev)
        with on_device(dev):
            _launch("pcb_nbr_mlp_backward_reduce", P * k * C * C, base.data_ptr(), rel.data_ptr(), P, k, C,
                    wr.data_ptr(), consts[0].data_ptr(), consts[1].data_ptr(), consts[2].data_ptr(),
                    consts[3].data_ptr(), w2.data_ptr(), g.data_ptr(), arg.data_ptr(), sums.data_ptr(),
                    dw2p.data_ptr())
            _launch("pcb_bn_bwd_finalize", C, sums.data_ptr(), parts, P * k, C, consts[0].data_ptr(),
                    consts[2].data_ptr(), consts[3].data_ptr(), training, pq[0].data_ptr(), pq[1].data_ptr(),
                    pq[2].data_ptr(), pq[3].data_ptr(), 0, 0)
            _launch("pcb_nbr_mlp_backward_apply", P * k * C * C, base.data_ptr(), rel.data_ptr(), P, k, C,
                    wr.data_ptr(), consts[0].data_ptr(), consts[1].data_ptr(), pq[0].data_ptr(), pq[1].data_ptr(),
                    w2.data_ptr(), g.data_ptr(), arg.data_ptr(), dbase.data_ptr(), dwrp.data_ptr())
        dw2 = sum_slabs(dw2p)
        return (dbase, None, sum_slabs(dwrp), pq[2].clone() if has_affine else None,
                pq[3].clone() if has_affine else None, dw2[:, :C].contiguous(),
                dw2[:, C].contiguous() if has_b2 else None, None, None, None, None, None, None)


def neighbour_mlp(base, rel, wr, bn, conv2):
    """max_j conv2(relu(bn(base[i] + wr . rel[i,j])))  ->  [P, C] fp32.
    base [P,C], rel [P,k,3] fp32, wr [C,3]; bn an nn.BatchNorm2d (bookkeeping as in its forward)."""
    momentum = bn.momentum if bn.momentum is not None else _bn_bookkeeping(bn)
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    track = bn.track_running_stats and bn.running_mean is not None
    C = conv2.out_channels
    return _NeighbourMLP.apply(
        base, rel, wr, bn.weight, bn.bias, conv2.weight.view(C, conv2.in_channels), conv2.bias,
        bn.running_mean if (track or not training) else None,
        bn.running_var if (track or not training) else None,
        training, momentum, bn.eps, _counter(bn))
