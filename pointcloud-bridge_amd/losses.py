"""Training criterion of the reference's BridgeSeg trainer (caller side of the hot path).

`BridgeStructureLoss`: models/model.py:169-260 of the reference, used by
train_MulSca_BriStruNet_CB.py:151-156,178 as `criterion(outputs, labels, points=points)`: a
label-smoothed cross entropy whose 5 class weights are raised, per step, where the PREDICTED
components violate the bridge's vertical order (abutment < girder < deck < parapet).  Same constructor,
buffer name and result.  The reference decides with Python `if tensor.any()` (one host sync per
class pair, ~20 per step); here the same conditions multiply the additions as 0/1 tensors, so the
criterion enqueues without reading anything back.  Plain torch ops on [B,N] masks -- no kernel of its
own; runs wherever its inputs live.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

# (class, classes that must lie below it, classes that must lie above it) -- models/model.py:176-182
_ORDER = ((1, (), (2, 3, 4)), (2, (1,), (3, 4)), (3, (1, 2), (4,)), (4, (1, 2, 3), ()))


class BridgeStructureLoss(nn.Module):
    def __init__(self, num_classes=5, alpha=20.0, rel_margin=0.2, class_weights=None):
        super().__init__()
        self.alpha = alpha
        self.rel_margin = rel_margin
        default_weights = torch.tensor([1.5, 1.0, 1.2, 1.5, 1.0])
        self.base_weights = default_weights if class_weights is None else class_weights
        self.register_buffer('base_weights_buffer', self.base_weights)

    @staticmethod
    def _mean_relative_height(points, mask):
        """models/model.py:189-196: z of the masked points, normalised by the extent of the MASKED
        cloud (unselected points count as the origin), averaged over the selected points -> [B]."""
        m = mask.to(points.dtype)
        masked = points * m.unsqueeze(-1)
        lo = masked.amin(dim=1, keepdim=True)
        hi = masked.amax(dim=1, keepdim=True)
        rel = (masked - lo) / (hi - lo + 1e-7)
        return (rel[..., 2] * m).sum(dim=1) / m.sum(dim=1).clamp(min=1)

    def forward(self, outputs, labels, points):
        """outputs [B,5,N] logits, labels [B,N] int64, points [B,N,3] -> scalar loss."""
        rows = _as_rows(outputs, False) if outputs.is_cuda else None
        if (rows is not None and rows.shape[1] == 5 and labels.is_cuda and labels.device == outputs.device
                and labels.dtype == torch.int64 and points.is_cuda and points.device == outputs.device
                and labels.numel() == rows.shape[0] and labels.shape[0] <= 1024):
            # GPU: two kernels for the class weights, the weighted + smoothed cross entropy on the logits rows in place
            # (csrc/loss.hip) -- no ATen reduction: the criterion may sit inside a captured step
            B, N = labels.shape
            w = _bridge_weights(rows, labels.reshape(-1).contiguous(), points.float().contiguous(), B, N, float(self.alpha),
                                float(self.rel_margin), self.base_weights_buffer.to(outputs.device, torch.float32))
            return _CrossEntropyRowsW.apply(rows, labels.reshape(-1).contiguous(), w, 0.2, -100)
        logits = outputs.transpose(1, 2)
        B = labels.shape[0]
        with torch.no_grad():
            preds = torch.argmax(logits, dim=-1)
            weights = self.base_weights_buffer.to(logits.device).repeat(B, 1)
            # a class takes part only if SOME scene of the batch carries it in the labels (:208-211,
            # :230, :240) -- a 0/1 scalar instead of an `if`
            present = {c: (labels == c).any().to(weights.dtype) for c in (1, 2, 3, 4)}
            height = {c: self._mean_relative_height(points, preds == c) for c in (1, 2, 3, 4)}  # :213-216
            cols = [weights[:, c] for c in range(weights.shape[1])]
            for cid, lower_classes, upper_classes in _ORDER:
                for low in lower_classes:                                   # 'above' entries, :228-236
                    v = F.relu(self.rel_margin - (height[cid] - height[low])) * present[low]
                    cols[cid] = cols[cid] + self.alpha * v
                    cols[low] = cols[low] + self.alpha * v * 0.5
                for up in upper_classes:                                    # 'below' entries, :238-246
                    v = F.relu(self.rel_margin - (height[up] - height[cid])) * present[up]
                    cols[cid] = cols[cid] + self.alpha * v
                    cols[up] = cols[up] + self.alpha * v * 0.3
            cols[0] = cols[0] + self.alpha * (1 - (preds == 0).to(weights.dtype).mean(dim=1))  # :250-251
            weights = torch.stack(cols, dim=1)
            freq = torch.bincount(labels.reshape(-1), minlength=5).to(weights.dtype).clamp(min=1)  # :253
            class_weights = 1 / freq.sqrt()
            class_weights = class_weights * class_weights.new_tensor([1.0, 2.0, 1.0, 1.0, 2.0])  # :255-256
            w = weights.mean(dim=0) * class_weights
        return F.cross_entropy(logits.reshape(-1, 5), labels.reshape(-1), weight=w, label_smoothing=0.2)


class _CrossEntropyRows(torch.autograd.Function):
    """Mean cross entropy over logits rows [R, C] fp32 (`ld` floats apart) against labels [R] -- csrc/loss.hip:
    one pass forward, one backward, instead of ATen's layout copy + log-softmax + nll_loss2d + fills."""

    @staticmethod
    def forward(ctx, rows, labels, ignore_index):
        from .ops import _launch, on_device
        R, C = rows.shape
        dev = rows.device
        lib = _lib.load()
        partials = torch.empty(2 * lib.pcb_cross_entropy_partials(R), dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        with on_device(dev):
            _launch("pcb_cross_entropy_fwd", R * C, rows.data_ptr(), rows.stride(0), labels.data_ptr(), R, C,
                    int(ignore_index), partials.data_ptr(), out.data_ptr())
        ctx.save_for_backward(rows, labels, out)
        ctx.ignore_index = int(ignore_index)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        from .ops import _launch, on_device
        rows, labels, out = ctx.saved_tensors
        R, C = rows.shape
        d = torch.empty(R, C, dtype=torch.float32, device=rows.device)
        g = g.to(torch.float32).contiguous()
        with on_device(rows.device):
            _launch("pcb_cross_entropy_bwd", R * C, rows.data_ptr(), rows.stride(0), labels.data_ptr(), R, C,
                    ctx.ignore_index, out.data_ptr(), g.data_ptr(), d.data_ptr())
        return d, None, None


def _bridge_weights(rows, labels, points, B, N, alpha, rel_margin, base):
    from .ops import _launch, on_device
    dev = rows.device
    stats = torch.empty(B, 22, dtype=torch.float32, device=dev)
    w = torch.empty(5, dtype=torch.float32, device=dev)
    with on_device(dev):
        _launch("pcb_bridge_loss_weights", B * N * 5, rows.data_ptr(), rows.stride(0), labels.data_ptr(), points.data_ptr(), B, N,
                alpha, rel_margin, base.contiguous().data_ptr(), stats.data_ptr(), w.data_ptr())
    return w


class _CrossEntropyRowsW(torch.autograd.Function):
    """F.cross_entropy(rows, labels, weight=w, label_smoothing=eps) over logits rows [R, C] fp32 (csrc/loss.hip); the
    class weights are data (no gradient), as in the reference (computed under no_grad, models/model.py:203)."""

    @staticmethod
    def forward(ctx, rows, labels, w, eps, ignore_index):
        from .ops import _launch, on_device
        R, C = rows.shape
        dev = rows.device
        partials = torch.empty(2 * _lib.load().pcb_cross_entropy_partials(R), dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        with on_device(dev):
            _launch("pcb_cross_entropy_w_fwd", R * C, rows.data_ptr(), rows.stride(0), labels.data_ptr(), R, C, int(ignore_index),
                    w.data_ptr(), float(eps), partials.data_ptr(), out.data_ptr())
        ctx.save_for_backward(rows, labels, w, out)
        ctx.cfg = (float(eps), int(ignore_index))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        from .ops import _launch, on_device
        rows, labels, w, out = ctx.saved_tensors
        eps, ignore_index = ctx.cfg
        R, C = rows.shape
        d = torch.empty(R, C, dtype=torch.float32, device=rows.device)
        g = g.to(torch.float32).contiguous()
        with on_device(rows.device):
            _launch("pcb_cross_entropy_w_bwd", R * C, rows.data_ptr(), rows.stride(0), labels.data_ptr(), R, C, ignore_index,
                    w.data_ptr(), eps, out.data_ptr(), g.data_ptr(), d.data_ptr())
        return d, None, None, None, None


def _as_rows(logits, channels_last):
    """The [R, C] fp32 row view behind the logits if they ARE rows in memory (what the networks of this package
    return: `rows.view(B, N, C).transpose(1, 2)` for the PointNet++ family, [B, N, C] for DGCNN); else None."""
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() in (2, 3)):
        return None
    x = logits if (channels_last or logits.dim() == 2) else logits.transpose(1, 2)  # -> [B, N, C]
    C = x.shape[-1]
    if C > 64 or x.stride(-1) != 1:
        return None
    if x.dim() == 3:
        if x.stride(0) != x.shape[1] * x.stride(1):
            return None
        x = x.reshape(-1, C)  # a view of the rows (their gradient is then rows as well: no layout copy in backward)
    return x if x.stride(0) >= C else None


_CHECK_LABELS = os.environ.get("PCB_CHECK_LABELS", "0") != "0"


def cross_entropy(logits, labels, channels_last=False, ignore_index=-100, check_labels=None):
    """nn.CrossEntropyLoss()(logits, labels) as the reference's trainers call it -- logits [B,C,N] (or
    [B,N,C] / [R,C] with channels_last), labels [B,N] / [R] int64 -- on the library's one-pass kernel when the
    logits are GPU fp32 rows in memory AND the labels live on the same GPU; any other input (CPU tensors or labels,
    labels on another device, other layouts or dtypes) goes to F.cross_entropy, the reference's own call, which
    raises its usual errors for mismatched devices.

    Labels outside [0, C) that are not ignore_index: F.cross_entropy trips a device-side assertion; the kernel
    (csrc/loss.hip) never dereferences them and counts such points as ignored -- bad label data would train
    silently.  check_labels=True (default: the environment variable PCB_CHECK_LABELS, off) validates the range first
    and raises IndexError; it costs a host synchronisation per call, so it is a debugging switch."""
    rows = _as_rows(logits, channels_last)
    same_gpu = labels.is_cuda and labels.device == logits.device
    if rows is None or not same_gpu or labels.dtype != torch.int64 or labels.numel() != rows.shape[0]:
        if channels_last and logits.dim() == 3:
            return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=ignore_index)
        return F.cross_entropy(logits, labels, ignore_index=ignore_index)
    if _CHECK_LABELS if check_labels is None else check_labels:
        bad = (labels != ignore_index) & ((labels < 0) | (labels >= rows.shape[1]))
        if bool(bad.any()):
            raise IndexError(f"cross_entropy: {int(bad.sum())} labels outside [0, {rows.shape[1]}) and != ignore_index")
    return _CrossEntropyRows.apply(rows, labels.reshape(-1).contiguous(), ignore_index)
