"""CPU oracle of the segmentation losses on the hot path, plain PyTorch fp32.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench).

Parity status: PINNED - ``tests/golden/make_golden.py losses`` imports the reference's own loss classes (biapy/engine/metrics.py,
with the five third-party metric imports of its header stubbed: they are never touched by these classes) and commits their values
and gradients on seeded inputs (``tests/golden/losses_golden.npz``); ``tests/test_oracle_golden.py`` checks this file against them.

Restates (paths relative to /root/reference):
  * bce .................. biapy/engine/metrics.py:493-586  CrossEntropyLoss_wrapper, num_classes <= 2 -> BCEWithLogitsLoss (mean)
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


def dice(logits: torch.Tensor, target: torch.Tensor, smooth: float = 1e-5) -> torch.Tensor:
    p = torch.sigmoid(logits)
    axes = [0] + list(range(2, logits.dim()))
    inter = torch.sum(p * target.float(), dim=axes)
    union = torch.sum(p, dim=axes) + torch.sum(target.float(), dim=axes)
    return 1.0 - torch.mean((2.0 * inter + smooth) / (union + smooth))


def dice_ce(logits: torch.Tensor, target: torch.Tensor, w_ce: float = 1.0, w_dice: float = 1.0, smooth: float = 1e-5) -> torch.Tensor:
    return w_ce * bce(logits, target) + w_dice * dice(logits, target, smooth)


def apply_head_activations(logits: torch.Tensor, acts: Sequence[str], training: bool = True) -> torch.Tensor:
    """base_workflow.py:1403-1457 for plain channels: ce_* activations are skipped in training, the others are applied."""
    outs = []
    for i, a in enumerate(acts):
        a = a.lower()
        x = logits[:, i:i + 1]
        if a == "linear" or (training and a in ("ce_sigmoid", "ce_softmax")):
            outs.append(x)
        elif a in ("ce_sigmoid", "sigmoid"):
            outs.append(torch.sigmoid(x))
        elif a == "tanh":
            outs.append(torch.tanh(x))
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
