"""Narrow 1x1 convolutions on fp32 rows through csrc/tinylin.hip (3..64 channels on B*N rows).

Reference layers: the Conv1d / Conv2d 1x1 of the bridge encoders, models/attention_modules.py:548-553,
:696-716, :759-764.  y = x W^T + b, its input gradient and its weight / bias gradient (per-block slabs
summed here).  GPU only, fp32.
"""
import torch

from . import _lib
from .ops import _launch, on_device

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
                slabs = torch.empty(parts, Co, Ci + 1, dtype=torch.float32, device=x.device)
                _launch("pcb_rows_linear_wgrad_f32", P * Ci * Co, g.data_ptr(), x.data_ptr(), P, Ci, Co, slabs.data_ptr())
                total = slabs.sum(dim=0)
                dw = total[:, :Ci].contiguous()
                db = total[:, Ci].contiguous() if ctx.has_bias else None
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
