"""CPU oracle for test-time augmentation (scalar-field predictions).

TEST INFRASTRUCTURE ONLY (checker for ``tests/``; nothing under ``biapy_amd/`` imports it).

Restated reference code (paths relative to /root/reference):
  * orientation group / transforms ... biapy/data/post_processing/tta.py:64-256 (AxisTransform.apply / .inverse,
                                        build_axis_transform_group)
  * padding, predict, undo, reduce ... biapy/data/post_processing/post_processing.py:1285-1383, 1386-1540
                                        (_pad_for_orientations, _crop_padding, _reduce_orientations, ensemble_predictions
                                        with tta_spec=None)

Parity status: PINNED.  The orientation group and the transforms against tta.py itself (tests/golden/tta_golden.npz); the pad /
predict / undo / reduce / crop pipeline against ``ensemble_predictions`` of post_processing.py run in the build container
(tests/golden/tta_ensemble_golden.npz, made by ``make_golden.py tta_ensemble``: the module imports once the third-party packages it
never calls on this path - cv2, fill_voids, scikit-image, numba ... - are replaced by stand-ins, tests/golden/_ref_shim.py).
"""
from __future__ import annotations

import itertools
import math

import numpy as np


def group(ndim: int, level: str = "full"):
    ident = tuple(range(ndim))
    if level == "none":
        return [(ident, (1,) * ndim)]
    inter = (0, 1) if ndim == 2 else (1, 2)
    perms = [ident] if level == "flips" else []
    if level == "full":
        for sub in itertools.permutations(inter):
            p = list(range(ndim))
            for slot, src in zip(inter, sub):
                p[slot] = src
            perms.append(tuple(p))
    out = [(p, s) for s in itertools.product((1, -1), repeat=ndim) for p in perms]
    out.sort(key=lambda t: not (t[0] == ident and all(v == 1 for v in t[1])))
    return out


def apply(arr: np.ndarray, perm, sign) -> np.ndarray:
    n = len(perm)
    out = np.transpose(arr, tuple(perm) + (n,))
    fl = tuple(a for a in range(n) if sign[a] < 0)
    if fl:
        out = np.flip(out, axis=fl)
    return np.ascontiguousarray(out)


def inverse(perm, sign):
    n = len(perm)
    pinv = [0] * n
    for a, p in enumerate(perm):
        pinv[p] = a
    return tuple(pinv), tuple(sign[pinv[b]] for b in range(n))


def ensemble(img: np.ndarray, pred_func, ndim: int, mode: str = "mean", level: str = "full", batch_size_value: int = 1) -> np.ndarray:
    ors = group(ndim, level)
    moved = set()
    for p, _ in ors:
        for a in range(ndim):
            if p[a] != a:
                moved.update((a, p[a]))
    pad_before = None
    if moved:
        target = max(img.shape[a] for a in moved)
        if not all(img.shape[a] == target for a in moved):
            pad_before = [0] * ndim
            for a in moved:
                pad_before[a] = target - img.shape[a]
            pm = "edge" if any(pad_before[a] >= img.shape[a] for a in moved) else "reflect"
            img = np.pad(img, [(pad_before[a], 0) for a in range(ndim)] + [(0, 0)], mode=pm)
    aug = np.stack([apply(img, p, s) for p, s in ors], 0)
    preds = []
    for i in range(int(math.ceil(aug.shape[0] / batch_size_value))):
        preds.append(pred_func(aug[i * batch_size_value:(i + 1) * batch_size_value]))
    pred = np.concatenate(preds, 0).astype(np.float32)
    for n, (p, s) in enumerate(ors):
        pred[n] = apply(pred[n], *inverse(p, s))
    out = np.mean(pred, axis=0) if mode == "mean" else (np.min if mode == "min" else np.max)(pred, axis=0)
    if pad_before is not None:
        out = out[tuple(slice(q, None) for q in pad_before) + (slice(None),)]
    return out


def standin_pred(batch: np.ndarray) -> np.ndarray:
    """A predictor that is NOT equivariant (it depends on the position inside the oriented patch), two output channels, exact float32
    products and sums only - the same function produced the fixture through the reference and serves the tests.
    batch: (n, spatial..., C) -> (n, spatial..., 2)."""
    ramp = np.linspace(0.0, 1.0, batch[0, ..., 0].size, dtype=np.float32).reshape(batch.shape[1:-1])
    return np.stack([np.stack([batch[k, ..., 0] * ramp + np.float32(0.25) * batch[k, ..., -1], batch[k, ..., 0] * batch[k, ..., 0] - ramp], -1)
                     for k in range(batch.shape[0])], 0).astype(np.float32)


ENSEMBLE_CASES = [  # (name, shape (spatial..., C), ndim)
    ("cube", (6, 8, 8, 1), 3), ("reflect_pad", (5, 6, 9, 2), 3), ("edge_pad", (4, 3, 9, 1), 3), ("plane", (7, 10, 1), 2), ("square", (12, 12, 3), 2),
]
ENSEMBLE_SETTINGS = [("mean", "full", 3), ("min", "full", 1), ("max", "flips", 2), ("mean", "flips", 16), ("max", "full", 5)]
