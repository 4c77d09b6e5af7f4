"""CPU oracle of the segmentation losses on the hot path, plain PyTorch fp32.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench).

Parity status: PINNED - ``tests/golden/make_golden.py losses`` imports the reference's own loss classes (biapy/engine/metrics.py,
with the five third-party metric imports of its header stubbed: they are never touched by these classes) and commits their values
and gradients on seeded inputs (``tests/golden/losses_golden.npz``); ``tests/test_oracle_golden.py`` checks this file against them.

Restates (paths relative to /root/reference):
  * bce .................. biapy/engine/metrics.py:493-586  CrossEntropyLoss_wrapper, num_classes <= 2 -> BCEWithLogitsLoss (mean)
  * softmax_ce ........... :493-586  CrossEntropyLoss_wrapper, num_classes > 2 -> torch.nn.CrossEntropyLoss(ignore_index, weight) on the label map
                           y_true[:, 0]: sum_v w[y_v] (logsumexp(z_v) - z_v[y_v]) / sum_v w[y_v] over the voxels with y_v != ignore_index; a LIST of
                           predictions (:566-583) is weighted 0.5^i / sum with the target rescaled by nearest-neighbour interpolation (:437-455)
  * dice ................. :726-762  DiceLoss: sigmoid, batch_dice sums over batch + space, 1 - mean_c (2I + s) / (U + s)
  * dice_ce .............. :764-973  DiceCELoss binary case: w_ce * BCEWithLogits + w_dice * dice
  * instance_channels .... :1418-1810 instance_segmentation_loss for plain channels without masks / class re-balancing / border
                           weights: sum_i w_i * mean(crit_i(pred[:, i], target[:, i])) with crit in {bce (on logits), mse, l1}; the
                           workflow applies the head activation (tanh for the 'D' channel, biapy/engine/instance_seg.py:405-409)
                           BEFORE the loss also in training (base_workflow.py:1427: only ce_* activations are skipped).
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn.functional as F


def bce(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return F.binary_cross_entropy_with_logits(logits, target.float())


def softmax_ce(logits: torch.Tensor, target: torch.Tensor, weight=None, ignore_index: int = -100) -> torch.Tensor:
    """logits (N, C, *space), target (N, 1, *space) class ids; written out (not F.cross_entropy) so that the fixture checks the formula."""
    C = logits.shape[1]
    y = target[:, 0].long()
    keep = y != ignore_index
    lse = torch.logsumexp(logits, dim=1)
    zy = torch.gather(logits, 1, torch.where(keep, y, torch.zeros_like(y)).unsqueeze(1))[:, 0]
    w = torch.ones(C, dtype=logits.dtype) if weight is None else torch.as_tensor(weight, dtype=logits.dtype)
    wy = torch.where(keep, w[torch.where(keep, y, torch.zeros_like(y))], torch.zeros((), dtype=logits.dtype))
    return torch.sum(wy * (lse - zy)) / torch.sum(wy)


def softmax_ce_deep(preds: Sequence[torch.Tensor], target: torch.Tensor, weight=None, ignore_index: int = -100, gamma: float = 0.5) -> torch.Tensor:
    ws = [gamma ** i for i in range(len(preds))]
    loss = 0
    for pd, wj in zip(preds, ws):
        yt = target if pd.shape[2:] == target.shape[2:] else F.interpolate(target.clone().float(), size=pd.shape[2:], mode="nearest")
        loss = loss + softmax_ce(pd, yt, weight, ignore_index) * (wj / sum(ws))
    return loss


def confusion_counts(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """(3, C): |P_c & T_c|, |P_c|, |T_c| of the argmax prediction against the label map without the ignored voxels (the counts behind the multi-class
    jaccard_index, metrics.py:170-176; torchmetrics itself is not installed here, so the IoU built on them is parity-unpinned)."""
    C = logits.shape[1]
    y = target[:, 0].long().reshape(-1)
    p = logits.argmax(1).reshape(-1)
    keep = y != ignore_index
    y, p = y[keep], p[keep]
    cm = torch.bincount(y * C + p, minlength=C * C).view(C, C).double()      # rows = label, columns = prediction
    return torch.stack([cm.diag(), cm.sum(0), cm.sum(1)])


def dice(logits: torch.Tensor, target: torch.Tensor, smooth: float = 1e-5, batch_dice: bool = True) -> torch.Tensor:
    p = torch.sigmoid(logits)
    axes = ([0] if batch_dice else []) + list(range(2, logits.dim()))     # :749-751: batch_dice=False keeps the batch axis, the mean runs over it
    inter = torch.sum(p * target.float(), dim=axes)
    union = torch.sum(p, dim=axes) + torch.sum(target.float(), dim=axes)
    return 1.0 - torch.mean((2.0 * inter + smooth) / (union + smooth))


def dice_ce(logits: torch.Tensor, target: torch.Tensor, w_ce: float = 1.0, w_dice: float = 1.0, smooth: float = 1e-5) -> torch.Tensor:
    return w_ce * bce(logits, target) + w_dice * dice(logits, target, smooth)


def apply_head_activations(logits: torch.Tensor, acts: Sequence[str], training: bool = True) -> torch.Tensor:
    """base_workflow.py:1367-1470 (``apply_model_activations``) for one activation name per channel: ``linear`` and - in training - the
    ``ce_*`` ones leave the channel alone, consecutive ``ce_softmax`` channels are ONE softmax group, the rest is applied per channel.
    (For "class" blocks the reference applies the block's first activation to the whole block; with one repeated name per block - what
    the configuration produces - that is the same thing.)  Pinned: tests/golden/head_acts_golden.npz, generated by the reference method."""
    outs, i, n = [], 0, len(acts)
    while i < n:
        a = acts[i].lower()
        if a == "linear" or (training and a in ("ce_sigmoid", "ce_softmax")):
            outs.append(logits[:, i:i + 1])
            i += 1
        elif a in ("ce_softmax", "softmax"):
            j = i
            while j < n and acts[j].lower() == a:
                j += 1
            outs.append(torch.softmax(logits[:, i:j], dim=1))
            i = j
        elif a in ("ce_sigmoid", "sigmoid"):
            outs.append(torch.sigmoid(logits[:, i:i + 1]))
            i += 1
        elif a == "tanh":
            outs.append(torch.tanh(logits[:, i:i + 1]))
            i += 1
        else:
            raise ValueError(a)
    return torch.cat(outs, 1)


def instance_channels(pred: torch.Tensor, target: torch.Tensor, losses: Sequence[str], weights: Sequence[float]) -> torch.Tensor:
    """pred: the model output AFTER the training-time head activations; one plain channel per entry of ``losses``."""
    total = 0
    for i, (name, w) in enumerate(zip(losses, weights)):
        p, t = pred[:, i:i + 1].float(), target[:, i:i + 1].float()
        if name == "bce":
            lt = F.binary_cross_entropy_with_logits(p, t, reduction="none")
        elif name == "mse":
            lt = (p - t) ** 2
        elif name in ("l1", "mae"):
            lt = (p - t).abs()
        else:
            raise ValueError(name)
        total = total + w * (lt.sum() / lt.numel())
    return total
