"""Segmentation losses / metrics of the binary (1-channel) head on the device.

Mirrors ``biapy/engine/metrics.py``: ``CrossEntropyLoss_wrapper`` (:493-586; the default ``LOSS.TYPE = "CE"`` on one output
channel is ``BCEWithLogitsLoss``), ``DiceLoss`` (:726-762, ``batch_dice=True``, smooth 1e-5), ``DiceCELoss`` (:764-973,
binary case: ``w_ce * BCE + w_dice * (1 - Dice)``) and ``jaccard_index`` (:138-232, threshold 0.5).  One streaming HIP kernel
produces every sum the four need (``bpx_seg_loss_sums``); the backward is one more pass (``bpx_seg_loss_bwd``).  Round 6: the multi-class
case of ``CrossEntropyLoss_wrapper`` (``num_classes > 2``: softmax cross entropy over 3..8 class channels with ``ignore_index`` and the "manual"
class weights) and the confusion counts of the multi-class IoU run on the device too (``bpx_softmax_ce_*``); ``DiceCELoss`` stays binary.
"""
from __future__ import annotations

import torch

from . import _lib as L

lib = L.lib


def _sums_and_loss(logits: torch.Tensor, target: torch.Tensor, w_ce: float = 1.0, w_dice: float = 0.0, smooth: float = 1e-5):
    """({sum bce, sum p*t, sum p, sum t, |P&T|, |P|T|} as a float64 device tensor, the loss as a 0-d float32 device tensor): one streaming
    pass + one single-block kernel that sums the partial rows in a fixed order in double and forms the loss (no host sync, no PyTorch
    element-wise launches: round 2's profile showed ~25 of them per training step)."""
    n = logits.numel()
    nb = lib.bpx_seg_loss_blocks(n)
    part = torch.empty((nb, 6), dtype=torch.float32, device=logits.device)
    L.check(lib.bpx_seg_loss_sums(logits.data_ptr(), target.data_ptr(), n, part.data_ptr(), L.stream_ptr()))
    sums = torch.empty(6, dtype=torch.float64, device=logits.device)
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    L.check(lib.bpx_seg_loss_finish(part.data_ptr(), nb, n, w_ce, w_dice, smooth, sums.data_ptr(), loss.data_ptr(), L.stream_ptr()))
    return sums, loss


def _sums(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return _sums_and_loss(logits, target)[0]


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
        s, loss = _sums_and_loss(z, t, w_ce, w_dice, smooth)
        ctx.save_for_backward(z, t, s)
        ctx.cfg = (w_ce, w_dice, smooth, z.numel())
        return loss

    @staticmethod
    def backward(ctx, g):
        z, t, s = ctx.saved_tensors
        w_ce, w_dice, smooth, n = ctx.cfg
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        dz = torch.empty_like(z)
        # the three coefficients a = w_ce g / n, b = 2 w_dice g / (U + s), c = w_dice g (2 I + s) / (U + s)^2 are formed inside the kernel
        L.check(lib.bpx_seg_loss_bwd_fused(z.data_ptr(), t.data_ptr(), n, s.data_ptr(), g.data_ptr(), w_ce, w_dice, smooth, dz.data_ptr(), L.stream_ptr()))
        return dz, None, None, None, None


class BCEWithLogitsLoss(torch.nn.Module):
    """``CrossEntropyLoss_wrapper`` on a 1-channel head (metrics.py:543-544)."""

    def forward(self, logits, target):
        return _SegLossFn.apply(logits, target, 1.0, 0.0, 1e-5)


class DiceLoss(torch.nn.Module):
    """metrics.py:726-762, one-channel head: ``batch_dice=True`` (sums over batch and space) and ``batch_dice=False`` (Dice per sample, then the mean)."""

    def __init__(self, batch_dice: bool = True, smooth: float = 1e-5):
        super().__init__()
        self.batch_dice, self.smooth = bool(batch_dice), smooth

    def forward(self, logits, target):
        if self.batch_dice:
            return _SegLossFn.apply(logits, target, 0.0, 1.0, self.smooth)
        # batch_dice=False (:749-751): the sums stay per sample, the loss is 1 - mean_n dice_n = mean_n (1 - dice_n): the same fused passes once per
        # sample (contiguous slices of the one-channel logits), averaged
        return torch.stack([_SegLossFn.apply(logits[i:i + 1], target[i:i + 1], 0.0, 1.0, self.smooth) for i in range(logits.shape[0])]).mean()


class DiceCELoss(torch.nn.Module):
    """Binary case of metrics.py:764-973: ``w_ce * BCEWithLogits + w_dice * DiceLoss``."""

    def __init__(self, w_ce: float = 1.0, w_dice: float = 1.0, smooth: float = 1e-5):
        super().__init__()
        self.w_ce, self.w_dice, self.smooth = float(w_ce), float(w_dice), float(smooth)

    def forward(self, logits, target):
        return _SegLossFn.apply(logits, target, self.w_ce, self.w_dice, self.smooth)


@torch.no_grad()
def _jaccard_binary(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    z, t = _prep(logits, target)
    s = _sums(z, t)
    return (s[4] / torch.clamp(s[5], min=1.0)).to(torch.float32)


class jaccard_index:
    """``biapy.engine.metrics.jaccard_index`` (metrics.py:138-232): the reference builds the metric once - ``jaccard_index(num_classes=, device=, t=,
    model_source=, ndim=, ignore_index=)`` - and calls it with ``(y_pred, y_true)`` per batch; the same here, on the fused device passes.
    ``num_classes <= 2``: IoU of ``sigmoid(logits) > 0.5`` against ``target > 0.5`` on the one-channel head (thresholds other than 0.5 and an
    ignore value are refused: the reference leaves them to torchmetrics); ``num_classes > 2``: ``jaccard_index_multiclass`` (parity-unpinned, see
    there).  A dict prediction is read at ``"pred"``, a list of predictions is averaged with the target rescaled by nearest-neighbour interpolation
    (:219-231).  ``jaccard_index(logits, target)`` with two tensors keeps the functional form of rounds 2-5 (binary IoU, 0-d device tensor)."""

    def __new__(cls, *args, **kwargs):
        if len(args) >= 2 and isinstance(args[0], torch.Tensor):
            return _jaccard_binary(args[0], args[1])
        return super().__new__(cls)

    def __init__(self, num_classes: int, device=None, t: float = 0.5, model_source: str = "biapy", ndim: int = 2, ignore_index: int = -1):
        self.num_classes, self.device, self.t, self.model_source, self.ndim = int(num_classes), device, float(t), model_source, ndim
        self.ignore_index = ignore_index if ignore_index != -1 else None
        if self.num_classes <= 2 and (self.t != 0.5 or self.ignore_index is not None):
            raise NotImplementedError("biapy_amd.losses.jaccard_index: the binary IoU kernel thresholds at 0.5 and has no ignore value; use the reference metric")
        if model_source != "biapy":
            raise NotImplementedError("biapy_amd.losses.jaccard_index: only model_source='biapy' (one-channel binary head / class channels)")

    @torch.no_grad()
    def __call__(self, y_pred, y_true):
        pds = y_pred["pred"] if isinstance(y_pred, dict) and "pred" in y_pred else y_pred
        if not isinstance(pds, list):
            pds = [pds]
        iou = 0
        for pd in pds:
            yt = y_true
            if pd.shape[-self.ndim:] != y_true.shape[-self.ndim:]:
                yt = torch.nn.functional.interpolate(y_true.clone().float(), size=pd.shape[-self.ndim:], mode="nearest")
            if self.num_classes > 2:
                iou = iou + jaccard_index_multiclass(pd, yt, -100 if self.ignore_index is None else self.ignore_index)
            else:
                iou = iou + _jaccard_binary(pd, yt)
        return iou / len(pds)


@torch.no_grad()
def hard_dice(logits: torch.Tensor, target: torch.Tensor, smooth: float = 1e-5) -> torch.Tensor:
    """Dice of the binarised prediction (the parity metric of BASELINE.json): 2|P&T| / (|P| + |T|)."""
    z, t = _prep(logits, target)
    s = _sums(z, t)
    return ((2.0 * s[4] + smooth) / (s[4] + s[5] + smooth)).to(torch.float32)   # |P| + |T| = |P&T| + |P|T|


# ---------------------------------------------------------------------------------------------------------------------------
# multi-class semantic segmentation (MODEL.N_CLASSES > 2): softmax cross entropy + per-class confusion counts
# ---------------------------------------------------------------------------------------------------------------------------
def _prep_classes(logits: torch.Tensor, target: torch.Tensor):
    if logits.dim() < 3 or not 2 <= logits.shape[1] <= 8:
        raise NotImplementedError("biapy_amd.losses: the multi-class cross entropy takes 2..8 class channels; use the reference loss beyond that")
    if not logits.is_cuda:
        raise RuntimeError("biapy_amd.losses run on the MI355X only (logits are on %s); there is no CPU path" % logits.device)
    if target.dim() == logits.dim() - 1:
        target = target.unsqueeze(1)
    if target.shape[0] != logits.shape[0] or target.shape[1] != 1 or target.shape[2:] != logits.shape[2:]:
        raise ValueError(f"target shape {tuple(target.shape)} does not match logits {tuple(logits.shape)} (one label channel expected)")
    return logits.contiguous().to(torch.float32), target.contiguous().to(torch.float32)      # class ids as floats (exact below 2^24)


def _softmax_ce_sums(z: torch.Tensor, t: torch.Tensor, ignore_index: int, weight):
    N, C = z.shape[0], z.shape[1]
    vox = z.numel() // (N * C)
    nb, row = lib.bpx_softmax_ce_blocks(vox), lib.bpx_softmax_ce_row()
    part = torch.empty((N * nb, row), dtype=torch.float32, device=z.device)
    L.check(lib.bpx_softmax_ce_sums(z.data_ptr(), t.data_ptr(), N, C, vox, ignore_index, L.ptr(weight), part.data_ptr(), L.stream_ptr()))
    sums = torch.empty(row, dtype=torch.float64, device=z.device)
    loss = torch.empty((), dtype=torch.float32, device=z.device)
    L.check(lib.bpx_softmax_ce_finish(part.data_ptr(), N, vox, sums.data_ptr(), loss.data_ptr(), L.stream_ptr()))
    return sums, loss


class _SoftmaxCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ignore_index):
        z, t = _prep_classes(logits, target)
        s, loss = _softmax_ce_sums(z, t, ignore_index, weight)
        ctx.save_for_backward(z, t, s, weight if weight is not None else z.new_empty(0))
        ctx.cfg = (ignore_index, weight is not None, logits.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        z, t, s, w = ctx.saved_tensors
        ignore_index, has_w, dtype = ctx.cfg
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        N, C = z.shape[0], z.shape[1]
        dz = torch.empty_like(z)
        L.check(lib.bpx_softmax_ce_bwd(z.data_ptr(), t.data_ptr(), N, C, z.numel() // (N * C), ignore_index, w.data_ptr() if has_w else None,
                                       s.data_ptr(), g.data_ptr(), dz.data_ptr(), L.stream_ptr()))
        return dz.to(dtype), None, None, None


class CrossEntropyLoss_wrapper(torch.nn.Module):
    """``biapy.engine.metrics.CrossEntropyLoss_wrapper`` (metrics.py:493-586), same constructor: ``num_classes <= 2`` is ``BCEWithLogitsLoss`` on the
    one-channel head, ``num_classes > 2`` is ``torch.nn.CrossEntropyLoss(ignore_index, weight)`` on the class channels against the label map
    ``y_true[:, 0]`` - both as fused device passes.  ``class_rebalance="manual"`` passes ``class_weights``; ``ignore_index=-1`` means torch's
    default -100 (:535).  A dict prediction is read at ``"pred"``; a LIST of predictions (deep supervision, :566-583) is weighted by 0.5^i / sum
    with the target rescaled by nearest-neighbour interpolation, as the reference does."""

    def __init__(self, num_classes: int, ndim: int = 2, class_rebalance: str = "none", class_weights=(), ignore_index: int = -1, device=None):
        super().__init__()
        self.ndim, self.num_classes, self.class_rebalance = ndim, int(num_classes), class_rebalance
        self.ignore_index = ignore_index if ignore_index != -1 else -100
        self.gamma = 0.5
        self.class_weights = None
        if class_rebalance == "manual":
            self.class_weights = torch.tensor(list(class_weights), dtype=torch.float32, device=device)
        if self.num_classes <= 2 and self.class_weights is not None:
            raise NotImplementedError("biapy_amd.losses: class weights of the binary (BCEWithLogits) case are not built; use the reference loss")

    def _one(self, pd, y_true):
        if self.num_classes <= 2:
            return _SegLossFn.apply(pd, y_true, 1.0, 0.0, 1e-5)
        w = self.class_weights
        if w is not None:
            if w.numel() != pd.shape[1]:
                raise ValueError(f"{w.numel()} class weights for {pd.shape[1]} class channels")
            if w.device != pd.device:
                w = self.class_weights = w.to(pd.device)
        return _SoftmaxCEFn.apply(pd, y_true[:, 0:1], w, self.ignore_index)

    def forward(self, y_pred, y_true):
        pds = y_pred["pred"] if isinstance(y_pred, dict) and "pred" in y_pred else y_pred
        if not isinstance(pds, list):
            pds, ws = [pds], [1.0]
        else:
            ws = [self.gamma ** i for i in range(len(pds))]
            ws = [x / sum(ws) for x in ws]
        loss = 0
        for pd, wj in zip(pds, ws):
            yt = y_true
            if pd.shape[-self.ndim:] != y_true.shape[-self.ndim:]:
                yt = torch.nn.functional.interpolate(y_true.clone().float(), size=pd.shape[-self.ndim:], mode="nearest")     # scale_target, metrics.py:437-455
            loss = loss + self._one(pd, yt) * wj
        return loss


@torch.no_grad()
def class_confusion_counts(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """(3, C) float64 device tensor {|P_c & T_c|, |P_c|, |T_c|} of the argmax prediction against the label map, labels equal to ``ignore_index``
    left out - the confusion counts behind the multi-class ``jaccard_index`` (metrics.py:170-176)."""
    z, t = _prep_classes(logits, target)
    s, _ = _softmax_ce_sums(z, t, ignore_index, None)
    return s[2:].view(3, 8)[:, : z.shape[1]]


@torch.no_grad()
def jaccard_index_multiclass(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """Macro-averaged IoU over the classes that occur in the prediction or the labels: mean_c tp_c / (|P_c| + |T_c| - tp_c).  The reference hands this
    to ``torchmetrics.JaccardIndex(task="multiclass")`` (metrics.py:170-173), which is not installed in this image: the definition is torchmetrics'
    documented macro average and is PARITY-UNPINNED (the counts themselves are checked against a confusion matrix in the tests)."""
    c = class_confusion_counts(logits, target, ignore_index)
    union = c[1] + c[2] - c[0]
    seen = union > 0
    return ((c[0] / torch.clamp(union, min=1.0)) * seen).sum().div(torch.clamp(seen.sum(), min=1)).to(torch.float32)


# ---------------------------------------------------------------------------------------------------------------------------
# multi-channel heads (row X / cfg 4: instance segmentation with B, C, D channels)
# ---------------------------------------------------------------------------------------------------------------------------
_KIND = {"bce": 0, "mse": 1, "l1": 2, "mae": 2}
_ACT = {"linear": 0, "ce_sigmoid": 0, "ce_softmax": 0, "tanh": 1, "sigmoid": 2}


class _ChanLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, codes, weights):
        if not logits.is_cuda:
            raise RuntimeError("biapy_amd.losses run on the MI355X only (logits are on %s); there is no CPU path" % logits.device)
        z, t = logits.contiguous().to(torch.float32), target.contiguous().to(torch.float32)
        N, Cc = z.shape[:2]
        vox = z[0, 0].numel()
        nb = lib.bpx_chan_loss_blocks(vox)
        part = torch.empty((N, Cc, nb), dtype=torch.float32, device=z.device)
        L.check(lib.bpx_chan_loss_sums(z.data_ptr(), t.data_ptr(), N, Cc, vox, codes, part.data_ptr(), L.stream_ptr()))
        # mean of every channel's terms (metrics.py:1784-1788), weighted and summed, by one single-block kernel (fixed-order double sums)
        w32 = weights if (weights.dtype == torch.float32 and weights.is_contiguous()) else weights.to(torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        L.check(lib.bpx_chan_loss_finish(part.data_ptr(), N, Cc, vox, w32.data_ptr(), loss.data_ptr(), L.stream_ptr()))
        ctx.save_for_backward(z, t, w32)
        ctx.cfg = (codes, N, Cc, vox)
        return loss

    @staticmethod
    def backward(ctx, g):
        z, t, weights = ctx.saved_tensors
        codes, N, Cc, vox = ctx.cfg
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        dz = torch.empty_like(z)
        L.check(lib.bpx_chan_loss_bwd_fused(z.data_ptr(), t.data_ptr(), N, Cc, vox, codes, weights.data_ptr(), g.data_ptr(), dz.data_ptr(), L.stream_ptr()))
        return dz, None, None, None


class InstanceChannelsLoss(torch.nn.Module):
    """``instance_segmentation_loss`` (biapy/engine/metrics.py:1418-1810) for plain channels - e.g. ``out_channels=["B","C","D"]``,
    ``losses_to_use=["bce","bce","mse"]`` - as one fused pass each way, taking the model's RAW logits: the head activation the
    workflow applies before the loss in training (``head_activations``, base_workflow.py:1403-1457; 'D' -> tanh,
    instance_seg.py:405-409) is part of the kernel.  Masks (``mask_values``), class re-balancing, border weights ('We'),
    multi-width channels ('R', 'A', discretised 'Db') and the separate class head stay on the reference implementation."""

    def __init__(self, channel_weights=(1, 1), out_channels=("F", "C"), losses_to_use=(), head_activations=None, channel_extra_opts=None,
                 class_rebalance_within_channels: bool = False, separated_class_channel: bool = False, ignore_index: int = -1, **_unused):
        super().__init__()
        chans = [c for c in out_channels if c not in ("We", "I")]
        if len(chans) != len(out_channels) or separated_class_channel or class_rebalance_within_channels or ignore_index != -1:
            raise NotImplementedError("InstanceChannelsLoss: border weights, class heads, re-balancing and ignore_index stay on the reference loss")
        if any(c in ("R", "A", "E_offset", "E_sigma", "E_seediness") for c in chans) or any((channel_extra_opts or {}).get(c, {}).get("mask_values") for c in chans):
            raise NotImplementedError("InstanceChannelsLoss: multi-width channels and masked channels stay on the reference loss")
        if any(c in ("Gv", "Gh", "Gz") for c in chans):
            # the reference multiplies the flow TARGET by flow_target_scale (5 for cellpose / omnipose fields, metrics.py:235-246, :1700-1705)
            raise NotImplementedError("InstanceChannelsLoss: flow channels (Gv / Gh / Gz: scaled targets) stay on the reference loss")
        if "Db" in chans and (channel_extra_opts or {}).get("Db", {}).get("val_type", "norm") == "discretize":
            # 11 prediction channels against one index channel, cross-entropy (metrics.py:1679-1688)
            raise NotImplementedError("InstanceChannelsLoss: a discretised 'Db' channel stays on the reference loss")
        if len(losses_to_use) != len(chans) or len(channel_weights) != len(chans) or len(chans) > 8:
            raise ValueError("one loss and one weight per output channel (at most 8 channels)")
        acts = list(head_activations) if head_activations is not None else ["tanh" if c == "D" else "ce_sigmoid" for c in chans]
        codes = 0
        for i, (name, a) in enumerate(zip(losses_to_use, acts)):
            if name not in _KIND or a.lower() not in _ACT:
                raise NotImplementedError(f"loss {name!r} / head activation {a!r} is not implemented on the MI355X path")
            kind = _KIND[name]
            act = _ACT[a.lower()] if kind != 0 else 0
            if kind == 0 and a.lower() not in ("ce_sigmoid", "linear"):
                raise NotImplementedError("a BCE channel takes logits (head activation ce_sigmoid)")
            codes |= (kind | (act << 2)) << (4 * i)
        self.codes = codes
        self.fused_head_activations = [a.lower() for a in acts]          # train_engine: the graph path hands this loss the raw logits
        self.register_buffer("weights", torch.tensor([float(w) for w in channel_weights], dtype=torch.float32))

    def forward(self, logits, target):
        logits = logits["pred"] if isinstance(logits, dict) else logits
        if logits.shape != target.shape or logits.shape[1] != self.weights.numel():
            raise ValueError(f"logits {tuple(logits.shape)} / target {tuple(target.shape)} do not match the {self.weights.numel()} configured channels")
        return _ChanLossFn.apply(logits, target, self.codes, self.weights.to(logits.device))
