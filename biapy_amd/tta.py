"""Test-time augmentation on the device.

Mirrors ``biapy/data/post_processing/tta.py`` (``AxisTransform`` :64-196, ``build_axis_transform_group`` :197-256) and the
scalar-field case of ``ensemble_predictions`` (``biapy/data/post_processing/post_processing.py:1386-1540``; what semantic
segmentation uses - ``tta_spec=None``): predict the patch in every orientation (8 in 2D, 16 in 3D: flips x in-plane 90 degree
rotations, Z is never swapped), undo each orientation, reduce with mean / min / max.

The volume stays on the MI355X: one gather per orientation (``bpx_tta_orient``) and one fused un-orient + reduce pass
(``bpx_tta_accumulate``) instead of the reference's NumPy stack of 16 copies; the reduction runs in the reference's orientation
order, so the float32 mean is the same sequential sum ``np.mean(stack, axis=0)`` forms.
Direction-carrying channels (flows, rays, offsets: ``tta_spec``) are not handled here - NotImplementedError.
"""
from __future__ import annotations

import ctypes as C
import itertools
from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L

lib = L.lib
TTA_GROUPS = ("auto", "full", "flips", "none")
Orientation = Tuple[Tuple[int, ...], Tuple[int, ...]]  # (perm, sign)


def build_axis_transform_group(ndim: int, level: str = "full", interchangeable_axes: Sequence[int] = None) -> List[Orientation]:
    """(perm, sign) of every orientation, identity first, in the reference's order (tta.py:197-256)."""
    if ndim not in (2, 3):
        raise ValueError("ndim must be 2 or 3; got {}".format(ndim))
    if level not in ("full", "flips", "none"):
        raise ValueError("level must be one of 'full', 'flips', 'none'; got '{}'".format(level))
    ident = tuple(range(ndim))
    if level == "none":
        return [(ident, (1,) * ndim)]
    inter = tuple(sorted(interchangeable_axes if interchangeable_axes is not None else ((0, 1) if ndim == 2 else (1, 2))))
    if level == "flips":
        perms = [ident]
    else:
        perms = []
        for sub in itertools.permutations(inter):
            p = list(range(ndim))
            for slot, src in zip(inter, sub):
                p[slot] = src
            perms.append(tuple(p))
    out = [(p, signs) for signs in itertools.product((1, -1), repeat=ndim) for p in perms]
    out.sort(key=lambda t: not (t[0] == ident and all(s == 1 for s in t[1])))   # stable: identity first
    return out


def _c3(perm, sign):
    """2D orientations act on (Y, X): embed as a 3D one on (Z=1, Y, X)."""
    if len(perm) == 2:
        perm, sign = (0, perm[0] + 1, perm[1] + 1), (1,) + tuple(sign)
    return (C.c_int * 3)(*perm), (C.c_int * 3)(*sign), tuple(perm)


@torch.no_grad()
def ensemble_predictions(vol: torch.Tensor, pred_func: Callable[[torch.Tensor], torch.Tensor], ndim: int, batch_size_value: int = 1,
                         mode: str = "mean", tta_spec=None, group: str = "auto") -> torch.Tensor:
    """vol: (spatial..., C) float32 device tensor; ``pred_func`` maps a (n, spatial..., C) batch to (n, spatial..., C_out)
    predictions of the same spatial shape.  Returns the ensembled (spatial..., C_out) prediction."""
    assert mode in ["mean", "min", "max"], "Get unknown ensemble mode {}".format(mode)
    assert ndim in (2, 3), "ndim must be 2 or 3, got {}".format(ndim)
    assert group in TTA_GROUPS, "group must be one of {}, got '{}'".format(TTA_GROUPS, group)
    if tta_spec is not None:
        raise NotImplementedError("biapy_amd.tta handles scalar-field predictions (tta_spec=None); use the reference for flows / rays / offsets")
    if not vol.is_cuda:
        raise RuntimeError("biapy_amd.tta runs on the MI355X only (volume is on %s); there is no CPU path" % vol.device)
    if vol.dim() != ndim + 1:
        raise ValueError("Expected a {}D input (spatial..., channels); got shape {}".format(ndim, tuple(vol.shape)))
    orientations = build_axis_transform_group(ndim, level=("full" if group == "auto" else group))
    img = vol.to(torch.float32)
    # square off the axes that get swapped (post_processing.py:1285-1339): front padding, reflect (edge if too short)
    moved = sorted({a for p, _ in orientations for a in range(ndim) if p[a] != a} | {p[a] for p, _ in orientations for a in range(ndim) if p[a] != a})
    pad_before = None
    if moved:
        target = max(img.shape[a] for a in moved)
        if not all(img.shape[a] == target for a in moved):
            pad_before = [target - img.shape[a] if a in moved else 0 for a in range(ndim)]
            pmode = "replicate" if any(pad_before[a] >= img.shape[a] for a in moved) else "reflect"
            t = img.movedim(-1, 0).unsqueeze(0)                       # (1, C, spatial...)
            pads = []
            for a in reversed(range(ndim)):
                pads += [pad_before[a], 0]
            img = F.pad(t, pads, mode=pmode).squeeze(0).movedim(0, -1)
    img = img.contiguous()
    sp = tuple(img.shape[:ndim])
    Z, Y, X = (1,) + sp if ndim == 2 else sp
    Cin = img.shape[-1]
    st = L.stream_ptr()
    acc = None
    n_or = len(orientations)
    for b0 in range(0, n_or, batch_size_value):
        chunk = orientations[b0:b0 + batch_size_value]
        batch = []
        for perm, sign in chunk:
            cp, cs, p3 = _c3(perm, sign)
            ext = (Z, Y, X)
            o = torch.empty(tuple(ext[p3[a]] for a in range(3))[3 - ndim:] + (Cin,), dtype=torch.float32, device=img.device)
            L.check(lib.bpx_tta_orient(img.data_ptr(), Z, Y, X, Cin, cp, cs, o.data_ptr(), st))
            batch.append(o)
        pred = pred_func(torch.stack(batch, 0)).to(torch.float32).contiguous()
        if tuple(pred.shape[1:1 + ndim]) != sp:
            raise ValueError("TTA needs the prediction to keep the input's spatial shape to undo the augmentation; got {} for an input of {}".format(
                tuple(pred.shape[1:1 + ndim]), sp))
        Cout = pred.shape[-1]
        if acc is None:
            acc = torch.empty(sp + (Cout,), dtype=torch.float32, device=img.device)
        for q, (perm, sign) in enumerate(chunk):
            k = b0 + q
            cp, cs, _ = _c3(perm, sign)
            L.check(lib.bpx_tta_accumulate(pred[q].data_ptr(), Z, Y, X, Cout, cp, cs, {"mean": 0, "min": 1, "max": 2}[mode], 1 if k == 0 else 0,
                                           n_or if (mode == "mean" and k == n_or - 1) else 0, acc.data_ptr(), st))
    if pad_before is not None:
        acc = acc[tuple(slice(p, None) for p in pad_before) + (slice(None),)].contiguous()
    return acc
