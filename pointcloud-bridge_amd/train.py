"""Training / evaluation harness for the drop-in networks on synthetic clouds.

The reference's trainers (train_MulSca_PN2.py, train_DGCNN.py, train_MulSca_BriStruNet_CB.py) and
inference.py do not travel to the GPU box and need h5py/tensorboard/wandb; this module reproduces
the parts of their loops that touch the hot path, with the same hyper-parameters:

  * batch dict keys points / colors / labels                     train_MulSca_PN2.py:155-157
  * Adam(lr=1e-3, betas=(.9,.999), weight_decay=1e-4)            train_MulSca_PN2.py:125
  * ReduceLROnPlateau(mode='max', factor=0.1, patience=5) on val accuracy      :127, :235
  * CrossEntropyLoss on [B,C,N] logits (PointNet++)              :161
  * BridgeStructureLoss(logits, labels, points) for BridgeSeg    train_MulSca_BriStruNet_CB.py:151-178 (..losses)
    or on [B*N,C] after a reshape (DGCNN)                        train_DGCNN.py:177-197
  * metrics from a confusion matrix: IoU = diag / (row + col - diag + 1e-6), mIoU = nanmean,
    OA = trace / total                                           inference.py:814-855
    (accumulated with one bincount instead of the reference's per-point Python loop, :226-231)
"""
import torch

from . import losses, parallel, rowmlp


def synthetic_scenes(num_scenes, num_points, num_classes=5, seed=0, device="cpu"):
    """Learnable synthetic segmentation task: unit-ball clouds (normalised as
    utils/simpdataset.py:47-62), label = horizontal slab of the point, colour = noisy label hue."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(num_scenes, num_points, 3, generator=g)
    p = v / v.norm(dim=-1, keepdim=True) * torch.rand(num_scenes, num_points, 1, generator=g) ** (1 / 3)
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=-1).max(dim=1)[0].view(-1, 1, 1)
    labels = ((p[:, :, 2] + 1.0) * 0.5 * num_classes).long().clamp_(0, num_classes - 1)
    colors = (torch.rand(num_scenes, num_points, 3, generator=g) * 0.5
              + 0.5 * (labels.unsqueeze(-1).float() / num_classes)).clamp_(0, 1)
    return {"points": p.contiguous().to(device), "colors": colors.to(device), "labels": labels.to(device)}


def segmentation_loss(logits, labels, channels_last=False):
    """CrossEntropy on [B,C,N] logits (PointNet++ family) or, channels_last, on [B,N,C] (DGCNN)."""
    return losses.cross_entropy(logits, labels, channels_last)


def predictions(logits, channels_last=False):
    return logits.argmax(dim=2 if channels_last else 1)


def confusion_matrix(pred, target, num_classes):
    """[num_classes, num_classes] counts, rows = true class (inference.py:226-231 as one bincount)."""
    k = target.reshape(-1) * num_classes + pred.reshape(-1)
    return torch.bincount(k, minlength=num_classes * num_classes).view(num_classes, num_classes)


def per_scene_confusion(pred, target, num_classes):
    """[B, num_classes, num_classes] counts of every scene of a batch in one bincount -- the per-file
    matrices inference.py:186-227 fills point by point."""
    B = target.shape[0]
    scene = torch.arange(B, device=target.device).view(B, 1) * (num_classes * num_classes)
    k = scene + target.reshape(B, -1) * num_classes + pred.reshape(B, -1)
    return torch.bincount(k.reshape(-1), minlength=B * num_classes * num_classes).view(B, num_classes, num_classes)


def metrics_from_confusion(cm):
    """calculate_metrics, inference.py:814-855, on a [C,C] count matrix (rows = true class): every
    quotient carries the reference's +1e-6, so a class that occurs nowhere has IoU 0 (not NaN) and
    DOES enter the mean.  Keys: miou, oa, iou, macc, acc, precision, recall, f1."""
    cm = cm.double()
    diag = cm.diag()
    rows, cols, total = cm.sum(1), cm.sum(0), cm.sum()
    iou = diag / (rows + cols - diag + 1e-6)
    acc = diag / (rows + 1e-6)
    prec_c = diag / (cols + 1e-6)
    weights = rows / total
    precision = (prec_c * weights).sum()
    recall = (acc * weights).sum()          # recall per class == accuracy per class (:832, :836)
    f1 = 2 * precision * recall / (precision + recall + 1e-6)
    return {"miou": float(torch.nanmean(iou)), "oa": float(diag.sum() / total), "iou": iou.tolist(),
            "macc": float(torch.nanmean(acc)), "acc": acc.tolist(), "precision": float(precision),
            "recall": float(recall), "f1": float(f1)}


class Trainer:
    """One model, the reference's optimiser and scheduler, optional data parallelism."""

    def __init__(self, model, num_classes=5, lr=1e-3, weight_decay=1e-4, distributed=False, criterion=None):
        """criterion: None = CrossEntropy (train_MulSca_PN2.py:161, train_DGCNN.py:177-197), or a module
        called as criterion(logits, labels, points) like losses.BridgeStructureLoss
        (train_MulSca_BriStruNet_CB.py:151-156, :178)."""
        self.model = model
        rowmlp.attach_step_operands(model)   # the model's own operand set (rowmlp.StepOperands)
        self.criterion = criterion
        self.num_classes = num_classes
        self.channels_last = type(model).__name__ == "DGCNN"  # DGCNN returns [B,N,C] (DGCNN.py:170)
        self.bucket = parallel.FlatGradAllReduce(model.parameters()) if distributed else None
        self.opt = torch.optim.Adam(model.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay)
        self.sched = torch.optim.lr_scheduler.ReduceLROnPlateau(self.opt, mode="max", factor=0.1, patience=5)

    def train_step(self, batch):
        self.model.train()
        if self.bucket is not None:
            self.bucket.zero()
        else:
            self.opt.zero_grad(set_to_none=True)
        logits = self.model(batch["points"], batch["colors"])
        if self.criterion is not None:
            loss = self.criterion(logits, batch["labels"], batch["points"])
        else:
            loss = segmentation_loss(logits, batch["labels"], self.channels_last)
        loss.backward()
        if self.bucket is not None:
            self.bucket.reduce()
        self.opt.step()
        rowmlp.prepare_step(self.model)   # GEMM operands of every stack from the updated weights, one launch
        return loss.detach()

    @torch.no_grad()
    def evaluate(self, batches):
        self.model.eval()
        cm = None
        for batch in batches:
            logits = self.model(batch["points"], batch["colors"])
            c = confusion_matrix(predictions(logits, self.channels_last), batch["labels"], self.num_classes)
            cm = c if cm is None else cm + c
        m = metrics_from_confusion(cm)
        self.sched.step(m["oa"])  # the reference steps its plateau scheduler on validation accuracy
        return m
