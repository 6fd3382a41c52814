"""Narrow 1x1 convolutions on fp32 rows through csrc/tinylin.hip (3..64 channels on B*N rows).

Reference layers: the Conv1d / Conv2d 1x1 of the bridge encoders, models/attention_modules.py:548-553,
:696-716, :759-764.  y = x W^T + b, its input gradient and its weight / bias gradient (per-block slabs
summed here).  GPU only, fp32.
"""
import torch

from . import _lib
from .ops import _launch, on_device, sum_slabs

MAX_CHANNELS = 64


class _RowsLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        P, Ci = x.shape
        Co = w.shape[0]
        x, w = x.contiguous(), w.contiguous()
        y = torch.empty(P, Co, dtype=torch.float32, device=x.device)
        with on_device(x.device):
            _launch("pcb_rows_linear_f32", P * Ci * Co, x.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(),
                    P, Ci, Co, y.data_ptr())
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        P, Ci = x.shape
        Co = w.shape[0]
        g = g.contiguous().float()
        dx = dw = db = None
        with on_device(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty(P, Ci, dtype=torch.float32, device=x.device)
                _launch("pcb_rows_linear_dgrad_f32", P * Ci * Co, g.data_ptr(), w.data_ptr(), P, Ci, Co, dx.data_ptr())
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                parts = _lib.load().pcb_rows_linear_wgrad_partials(P)
                slabs = torch.empty(parts, Co * Ci + Co, dtype=torch.float32, device=x.device)
                _launch("pcb_rows_linear_wgrad_f32", P * Ci * Co, g.data_ptr(), x.data_ptr(), P, Ci, Co, slabs.data_ptr())
                total = sum_slabs(slabs)          # [Co*Ci | Co]: both gradients are contiguous views of it
                dw = total[:Co * Ci].view(Co, Ci)
                db = total[Co * Ci:] if ctx.has_bias else None
        return dx, dw, db


def rows_linear(x, w, b=None):
    """x [P,Ci] fp32 rows, w [Co,Ci], b [Co] or None -> [P,Co] fp32."""
    if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1]:
        raise ValueError(f"rows_linear: x {tuple(x.shape)} does not match w {tuple(w.shape)}")
    if max(w.shape) > MAX_CHANNELS:
        raise ValueError(f"rows_linear serves up to {MAX_CHANNELS} channels, got {tuple(w.shape)}")
    if not x.is_cuda:
        raise RuntimeError("pointcloud_bridge_amd operators run on the GPU only (HIP kernels, no CPU fallback)")
    return _RowsLinear.apply(x.float(), w, b)


ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


class _NarrowBNAct(torch.autograd.Function):
    """act(BatchNorm(x)) on fp32 rows [R, C] with ANY channel count up to 64 (csrc/narrowbn.hip): statistics slabs,
    pcb_bn_finalize (running statistics, num_batches_tracked as nn.BatchNorm keeps them), one apply pass; backward:
    slabs of the two sums, their totals, one apply pass.  Kernels only -- valid inside a captured step."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act, nbt):
        R, C = x.shape
        dev = x.device
        lib = _lib.load()
        consts = torch.empty(4, C, dtype=torch.float32, device=dev)   # scale | shift | mean | invstd
        out = torch.empty(R, C, dtype=torch.float32, device=dev)
        with on_device(dev):
            nparts, slabs = 1, None
            if training:
                nparts = lib.pcb_rows_bn_partials(R, C)
                slabs = torch.empty(nparts, 2, C, dtype=torch.float32, device=dev)
                _launch("pcb_rows_bn_stats_f32", R * C, x.data_ptr(), R, C, slabs.data_ptr(), nparts)
            _launch("pcb_bn_finalize", C, 0 if slabs is None else slabs.data_ptr(), nparts, R, 0, C,
                    0 if gamma is None else gamma.data_ptr(), 0 if beta is None else beta.data_ptr(), 0,
                    0 if running_mean is None else running_mean.data_ptr(),
                    0 if running_var is None else running_var.data_ptr(), float(momentum), float(eps), int(training),
                    consts[0].data_ptr(), consts[1].data_ptr(), consts[2].data_ptr(), consts[3].data_ptr(),
                    0 if nbt is None else nbt.data_ptr())
            _launch("pcb_rows_bn_act_f32", R * C, x.data_ptr(), consts[0].data_ptr(), consts[1].data_ptr(), R, C, act,
                    out.data_ptr())
        ctx.save_for_backward(x, consts)
        ctx.cfg = (int(training), act, gamma is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, consts = ctx.saved_tensors
        training, act, has_affine = ctx.cfg
        R, C = x.shape
        dev = x.device
        lib = _lib.load()
        g = g.contiguous().float()
        nparts = lib.pcb_rows_bn_partials(R, C)
        slabs = torch.empty(nparts, 2, C, dtype=torch.float32, device=dev)
        dx = torch.empty(R, C, dtype=torch.float32, device=dev)
        with on_device(dev):
            _launch("pcb_rows_bn_act_bwd_reduce_f32", R * C, g.data_ptr(), x.data_ptr(), consts[0].data_ptr(),
                    consts[1].data_ptr(), consts[2].data_ptr(), consts[3].data_ptr(), R, C, act, slabs.data_ptr(), nparts)
            sums = sum_slabs(slabs)
            _launch("pcb_rows_bn_act_bwd_apply_f32", R * C, g.data_ptr(), x.data_ptr(), consts[0].data_ptr(),
                    consts[1].data_ptr(), consts[2].data_ptr(), consts[3].data_ptr(), sums.data_ptr(), R, C, act, training,
                    dx.data_ptr())
        # (dgamma / dbeta as rows of the freshly summed [2, C] tensor: views, no copy launches)
        return (dx, sums[1] if has_affine else None, sums[0] if has_affine else None,
                None, None, None, None, None, None, None)


def bn_act_rows(bn, x, act=ACT_NONE):
    """act(bn(x)) for fp32 rows x [R, C], C <= 64 (any C), with bn's parameters, running statistics and counter kept
    exactly as nn.BatchNorm keeps them (models/attention_modules.py:696-716, :759-764)."""
    from . import rowmlp
    if not x.is_cuda:
        raise RuntimeError("pointcloud_bridge_amd operators run on the GPU only (HIP kernels, no CPU fallback)")
    momentum = bn.momentum if bn.momentum is not None else rowmlp._bn_bookkeeping(bn)
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    track = bn.track_running_stats and bn.running_mean is not None
    if training:
        rowmlp.note_parameter_update(weights=False)
    return _NarrowBNAct.apply(x.float().contiguous(), bn.weight, bn.bias,
                              bn.running_mean if (track or not training) else None,
                              bn.running_var if (track or not training) else None,
                              training, momentum, bn.eps, act, rowmlp._counter(bn))


class _SceneMean(torch.autograd.Function):
    """Mean over the N rows of every scene: x [B*N, C] -> [B, C] (AdaptiveAvgPool1d(1) of color_context,
    models/attention_modules.py:718-722), as slab sums without atomics (csrc/narrowbn.hip)."""

    @staticmethod
    def forward(ctx, x, B, N):
        C = x.shape[1]
        lib = _lib.load()
        nparts = lib.pcb_scene_sum_partials(N)
        slabs = torch.empty(nparts, B, C, dtype=torch.float32, device=x.device)
        with on_device(x.device):
            _launch("pcb_scene_sum_f32", B * N * C, x.data_ptr(), B, N, C, slabs.data_ptr(), nparts)
        ctx.cfg = (B, N, C)
        return sum_slabs(slabs) * (1.0 / N)

    @staticmethod
    def backward(ctx, g):
        B, N, C = ctx.cfg
        return (g * (1.0 / N)).view(B, 1, C).expand(B, N, C).reshape(B * N, C), None, None


def scene_mean(x, B, N):
    if x.shape[1] > MAX_CHANNELS or not x.is_cuda:
        return x.view(B, N, -1).mean(dim=1)
    return _SceneMean.apply(x.float().contiguous(), B, N)


class _SceneScale(torch.autograd.Function):
    """x [B*N, C] * ctx [B, C] broadcast over each scene's rows (the context gate of ColorFeatureExtraction,
    models/attention_modules.py:751).  Backward: the gate's gradient is a per-scene column sum of g*x -- slab sums
    (csrc/narrowbn.hip) instead of an ATen reduction over 16384 rows per output."""

    @staticmethod
    def forward(ctx_, x, gate, B, N):
        ctx_.save_for_backward(x, gate)
        ctx_.cfg = (B, N)
        return (x.view(B, N, -1) * gate.unsqueeze(1)).view(B * N, -1)

    @staticmethod
    def backward(ctx_, g):
        x, gate = ctx_.saved_tensors
        B, N = ctx_.cfg
        C = x.shape[1]
        g = g.contiguous()
        dx = (g.view(B, N, C) * gate.unsqueeze(1)).view(B * N, C) if ctx_.needs_input_grad[0] else None
        dgate = None
        if ctx_.needs_input_grad[1]:
            prod = (g * x).contiguous()
            nparts = _lib.load().pcb_scene_sum_partials(N)
            slabs = torch.empty(nparts, B, C, dtype=torch.float32, device=x.device)
            with on_device(x.device):
                _launch("pcb_scene_sum_f32", B * N * C, prod.data_ptr(), B, N, C, slabs.data_ptr(), nparts)
            dgate = sum_slabs(slabs)
        return dx, dgate, None, None


def scene_scale(x, gate, B, N):
    """x [B*N, C] fp32 rows times gate [B, C], every scene's rows by its own gate row."""
    if x.shape[1] > MAX_CHANNELS or not x.is_cuda:
        return (x.view(B, N, -1) * gate.unsqueeze(1)).view(B * N, -1)
    return _SceneScale.apply(x.float().contiguous(), gate.float().contiguous(), B, N)
