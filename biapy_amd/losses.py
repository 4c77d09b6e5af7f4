"""Segmentation losses / metrics of the binary (1-channel) head on the device.

Mirrors ``biapy/engine/metrics.py``: ``CrossEntropyLoss_wrapper`` (:493-586; the default ``LOSS.TYPE = "CE"`` on one output
channel is ``BCEWithLogitsLoss``), ``DiceLoss`` (:726-762, ``batch_dice=True``, smooth 1e-5), ``DiceCELoss`` (:764-973,
binary case: ``w_ce * BCE + w_dice * (1 - Dice)``) and ``jaccard_index`` (:138-232, threshold 0.5).  One streaming HIP kernel
produces every sum the four need (``bpx_seg_loss_sums``); the backward is one more pass (``bpx_seg_loss_bwd``).  Multi-class
heads / class re-balancing / ignore_index stay on the reference implementation (NotImplementedError here).
"""
from __future__ import annotations

import torch

from . import _lib as L

lib = L.lib


def _sums(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """{sum bce, sum p*t, sum p, sum t, |P&T|, |P|T|} as a float64 device tensor (no host sync)."""
    n = logits.numel()
    part = torch.empty((lib.bpx_seg_loss_blocks(n), 6), dtype=torch.float32, device=logits.device)
    L.check(lib.bpx_seg_loss_sums(logits.data_ptr(), target.data_ptr(), n, part.data_ptr(), L.stream_ptr()))
    return part.to(torch.float64).sum(0)


def _prep(logits: torch.Tensor, target: torch.Tensor):
    if not logits.is_cuda:
        raise RuntimeError("biapy_amd.losses run on the MI355X only (logits are on %s); there is no CPU path" % logits.device)
    if logits.dim() < 3 or logits.shape[1] != 1:
        raise NotImplementedError("biapy_amd.losses implement the 1-channel (binary) head; use the reference losses for multi-class outputs")
    if target.shape != logits.shape:
        raise ValueError(f"target shape {tuple(target.shape)} != logits shape {tuple(logits.shape)}")
    return logits.contiguous().to(torch.float32), target.contiguous().to(torch.float32)


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, w_ce, w_dice, smooth):
        z, t = _prep(logits, target)
        s = _sums(z, t)
        n = z.numel()
        inter, union = s[1], s[2] + s[3]
        loss = w_ce * s[0] / n + w_dice * (1.0 - (2.0 * inter + smooth) / (union + smooth))
        ctx.save_for_backward(z, t, s)
        ctx.cfg = (w_ce, w_dice, smooth, n)
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        z, t, s = ctx.saved_tensors
        w_ce, w_dice, smooth, n = ctx.cfg
        g = g.to(torch.float64)
        den = s[2] + s[3] + smooth
        coef = torch.stack([w_ce * g / n, 2.0 * w_dice * g / den, w_dice * g * (2.0 * s[1] + smooth) / (den * den)]).to(torch.float32)
        dz = torch.empty_like(z)
        L.check(lib.bpx_seg_loss_bwd(z.data_ptr(), t.data_ptr(), n, coef.data_ptr(), dz.data_ptr(), L.stream_ptr()))
        return dz, None, None, None, None


class BCEWithLogitsLoss(torch.nn.Module):
    """``CrossEntropyLoss_wrapper`` on a 1-channel head (metrics.py:543-544)."""

    def forward(self, logits, target):
        return _SegLossFn.apply(logits, target, 1.0, 0.0, 1e-5)


class DiceLoss(torch.nn.Module):
    """metrics.py:726-762 with ``batch_dice=True``."""

    def __init__(self, batch_dice: bool = True, smooth: float = 1e-5):
        super().__init__()
        if not batch_dice:
            raise NotImplementedError("per-sample Dice (batch_dice=False) is not implemented on the MI355X path")
        self.smooth = smooth

    def forward(self, logits, target):
        return _SegLossFn.apply(logits, target, 0.0, 1.0, self.smooth)


class DiceCELoss(torch.nn.Module):
    """Binary case of metrics.py:764-973: ``w_ce * BCEWithLogits + w_dice * DiceLoss``."""

    def __init__(self, w_ce: float = 1.0, w_dice: float = 1.0, smooth: float = 1e-5):
        super().__init__()
        self.w_ce, self.w_dice, self.smooth = float(w_ce), float(w_dice), float(smooth)

    def forward(self, logits, target):
        return _SegLossFn.apply(logits, target, self.w_ce, self.w_dice, self.smooth)


@torch.no_grad()
def jaccard_index(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """IoU of (sigmoid(logits) > 0.5) vs (target > 0.5) (metrics.py:138-232, binary case); 0-d device tensor."""
    z, t = _prep(logits, target)
    s = _sums(z, t)
    return (s[4] / torch.clamp(s[5], min=1.0)).to(torch.float32)


@torch.no_grad()
def hard_dice(logits: torch.Tensor, target: torch.Tensor, smooth: float = 1e-5) -> torch.Tensor:
    """Dice of the binarised prediction (the parity metric of BASELINE.json): 2|P&T| / (|P| + |T|)."""
    z, t = _prep(logits, target)
    s = _sums(z, t)
    return ((2.0 * s[4] + smooth) / (s[4] + s[5] + smooth)).to(torch.float32)   # |P| + |T| = |P&T| + |P|T|
