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


# ---------------------------------------------------------------------------------------------------------------------------------------
# Direction-carrying channels (round 3): biapy/data/post_processing/tta.py:270-640 (ChannelGroup.supports / .remap of ScalarChannels,
# VectorChannels, RayChannels, AffinityChannels; TTASpec.filter_orientations / .remap_channels / .mode_reducible_channels) and the
# spec-aware branch of ensemble_predictions (post_processing.py:1477-1530: filtered orientations, zero padding, remap after the spatial
# un-orient, min / max only on the mode-reducible channels).  A spec is restated as a list of plain dicts:
#   {"kind": "scalar", "channels": [...]}
#   {"kind": "vector", "axis_channels": [c or None per spatial axis], "signed": bool, "axis_scale": None or [float per axis]}
#   {"kind": "rays", "start": first channel, "dirs": (nrays, ndim) unit directions in spatial-axis order}
#   {"kind": "affinities", "layout": {(spatial axis, offset): channel}}
# Parity status: PINNED against the reference classes driven through ensemble_predictions (tests/golden/tta_spec_golden.npz,
# make_golden.py tta_spec).
# ---------------------------------------------------------------------------------------------------------------------------------------
def transform_vectors(vecs: np.ndarray, perm, sign) -> np.ndarray:
    out = np.empty_like(vecs)
    for a in range(len(perm)):
        out[..., a] = sign[a] * vecs[..., perm[a]]
    return out


def ray_permutation(dirs: np.ndarray, perm, sign):
    """dest[j] = k with dirs[k] == inverse(t)(dirs[j]), or None (tta.py:440-463)."""
    if len(dirs) == 0:
        return None
    target = transform_vectors(np.asarray(dirs), *inverse(perm, sign))
    dots = target @ np.asarray(dirs).T
    dest = np.argmax(dots, axis=1)
    if np.allclose(dots[np.arange(len(dest)), dest], 1.0, atol=1e-4) and len(np.unique(dest)) == len(dest):
        return dest.astype(np.int64)
    return None


def group_supports(g: dict, perm, sign) -> bool:
    n = len(perm)
    if g["kind"] == "vector":
        pinv, _ = inverse(perm, sign)
        ac, sc = g["axis_channels"], g.get("axis_scale")
        for a in range(n):
            src = pinv[a]
            if (ac[a] is None) != (ac[src] is None):
                return False
            if sc is not None and src != a and not np.isclose(sc[a], sc[src]):
                return False
        return True
    if g["kind"] == "rays":
        return len(g["dirs"]) == 0 or ray_permutation(g["dirs"], perm, sign) is not None
    if g["kind"] == "affinities":
        return all((perm[axis], off) in g["layout"] for (axis, off) in g["layout"])
    return True


def filter_orientations(groups, ors):
    kept = [(p, s) for p, s in ors if all(group_supports(g, p, s) for g in groups)]
    return kept or [(tuple(range(len(ors[0][0]))), (1,) * len(ors[0][0]))]


def remap_channels(groups, pred: np.ndarray, perm, sign) -> None:
    """In place, on a spatially restored prediction (spatial..., C)."""
    n = len(perm)
    if tuple(perm) == tuple(range(n)) and all(s == 1 for s in sign):
        return
    pinv, sinv = inverse(perm, sign)
    for g in groups:
        if g["kind"] == "vector":
            ac = g["axis_channels"]
            src = [pred[..., c].copy() if c is not None else None for c in ac]
            for a, dst in enumerate(ac):
                if dst is None:
                    continue
                comp = src[pinv[a]]
                pred[..., dst] = comp if (sinv[a] > 0 or not g.get("signed", True)) else -comp
        elif g["kind"] == "rays":
            dest = ray_permutation(g["dirs"], perm, sign)
            s0, nr = g["start"], len(g["dirs"])
            block = pred[..., s0:s0 + nr].copy()
            pred[..., s0 + dest] = block
        elif g["kind"] == "affinities":
            src = {key: pred[..., ch].copy() for key, ch in g["layout"].items()}
            for (axis, off), block in src.items():
                dst_axis = perm[axis]
                dst = g["layout"][(dst_axis, off)]
                if sign[axis] > 0:
                    pred[..., dst] = block
                else:   # the map shifted `off` voxels up the axis, its first slice repeated into the `off` slices without a source (tta.py:531-543)
                    n_b = block.shape[dst_axis]
                    if 0 < off < n_b:
                        shifted = np.take(block, np.maximum(np.arange(n_b) - off, 0), axis=dst_axis)
                    else:
                        shifted = np.roll(block, shift=off, axis=dst_axis)
                    pred[..., dst] = shifted


def mode_reducible_channels(groups):
    out = []
    for g in groups:
        if g["kind"] == "scalar":
            out += list(g["channels"])
        elif g["kind"] == "vector" and not g.get("signed", True):
            out += [c for c in g["axis_channels"] if c is not None]
        elif g["kind"] == "rays":
            out += list(range(g["start"], g["start"] + len(g["dirs"])))
        elif g["kind"] == "affinities":
            out += sorted(g["layout"].values())
    return sorted(out)


def ensemble_spec(img: np.ndarray, pred_func, ndim: int, groups, mode: str = "mean", level: str = "full", batch_size_value: int = 1) -> np.ndarray:
    ors = filter_orientations(groups, group(ndim, level))
    scalar_only = all(g["kind"] == "scalar" for g in groups)
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
            if scalar_only:
                pm = "edge" if any(pad_before[a] >= img.shape[a] for a in moved) else "reflect"
                img = np.pad(img, [(pad_before[a], 0) for a in range(ndim)] + [(0, 0)], mode=pm)
            else:
                img = np.pad(img, [(pad_before[a], 0) for a in range(ndim)] + [(0, 0)], mode="constant")
    aug = np.stack([apply(img, p, s) for p, s in ors], 0)
    preds = []
    for i in range(int(math.ceil(aug.shape[0] / batch_size_value))):
        preds.append(pred_func(aug[i * batch_size_value:(i + 1) * batch_size_value]))
    pred = np.concatenate(preds, 0).astype(np.float32)
    for n, (p, s) in enumerate(ors):
        r = apply(pred[n], *inverse(p, s))
        remap_channels(groups, r, p, s)
        pred[n] = r
    out = np.mean(pred, axis=0)
    if mode != "mean":
        f = np.min if mode == "min" else np.max
        idx = np.asarray(mode_reducible_channels(groups), dtype=np.int64)
        if len(idx):
            out[..., idx] = f(pred[..., idx], axis=0)
    if pad_before is not None:
        out = out[tuple(slice(q, None) for q in pad_before) + (slice(None),)]
    return out


def standin_pred_multi(batch: np.ndarray, cout: int) -> np.ndarray:
    """Position-dependent (NOT equivariant) predictor with ``cout`` channels out of exact float32 operations: channel c is
    x * (1 + c/8) * ramp + c/4 - ramp/(c+1) evaluated term by term in float32 - identical bits on the CPU and on the device."""
    ramp = np.linspace(0.0, 1.0, batch[0, ..., 0].size, dtype=np.float32).reshape(batch.shape[1:-1])
    chans = []
    for c in range(cout):
        x = batch[..., c % batch.shape[-1]]
        chans.append((x * np.float32(1.0 + c / 8.0)) * ramp[None] + np.float32(c / 4.0) - ramp[None] * np.float32(1.0 / (c + 1)))
    return np.stack(chans, -1).astype(np.float32)


def uniform_rays_2d(nrays: int) -> np.ndarray:
    """Unit ray directions of the 2-D uniform-angle grid in (y, x) order."""
    ang = 2.0 * np.pi * np.arange(nrays) / nrays
    return np.stack([np.sin(ang), np.cos(ang)], 1).astype(np.float32)


def spec_cases():
    """(name, image shape (spatial..., C), ndim, cout, groups)"""
    return [
        ("flows3d", (4, 6, 9, 1), 3, 8, [
            {"kind": "vector", "axis_channels": [0, 1, 2], "signed": True, "axis_scale": None},
            {"kind": "vector", "axis_channels": [3, 4, 5], "signed": False, "axis_scale": None},
            {"kind": "scalar", "channels": [6, 7]}]),
        ("flows2d_in_3d", (3, 8, 8, 2), 3, 3, [          # no z component: orientations are all supported (z is never permuted)
            {"kind": "vector", "axis_channels": [None, 0, 1], "signed": True, "axis_scale": None},
            {"kind": "scalar", "channels": [2]}]),
        ("aniso_offsets", (4, 6, 6, 1), 3, 4, [          # y and x on different physical scales: the y/x swaps are dropped
            {"kind": "vector", "axis_channels": [0, 1, 2], "signed": True, "axis_scale": [2.0, 1.0, 0.5]},
            {"kind": "scalar", "channels": [3]}]),
        ("rays2d", (7, 10, 1), 2, 9, [
            {"kind": "rays", "start": 0, "dirs": uniform_rays_2d(8)},
            {"kind": "scalar", "channels": [8]}]),
        ("affinities3d", (4, 7, 7, 1), 3, 6, [
            {"kind": "affinities", "layout": {(0, 1): 0, (1, 1): 1, (2, 1): 2, (1, 2): 3, (2, 2): 4}},
            {"kind": "scalar", "channels": [5]}]),
    ]


SPEC_SETTINGS = [("mean", "full", 3), ("max", "full", 2), ("min", "flips", 16)]


# (channel names, ndim, channel_extra_opts, anisotropy) handed to the reference's build_tta_spec by make_golden.py tta_spec; the product's
# restatement of that function (biapy_amd.tta.build_tta_spec: host logic) is compared with the recorded structures
BUILD_SPEC_CASES = [
    (["Gz", "Gv", "Gh", "B", "E_sigma_0", "E_sigma_1", "E_sigma_2"], 3, None, None),
    (["Gv", "Gh", "B"], 2, None, None),
    (["B", "C", "V", "H", "Z"], 2, None, None),                                        # a z component on 2-D data becomes a scalar
    (["E_offset_0", "E_offset_1", "E_offset_2", "E_sigma_0", "E_sigma_1", "E_sigma_2", "E_seediness"], 3, None, (2.0, 1.0, 1.0)),
    (["B"] + ["R_%d" % i for i in range(32)], 2, {"R": {"nrays": 32}}, None),
    (["R_%d" % i for i in range(16)] + ["C"], 3, None, None),
    (["Az_1", "Ay_1", "Ax_1", "Ay_2", "Ax_2", "F"], 3, None, None),
    (["Az_1", "Ay_1", "Ax_1", "D"], 2, None, None),                                    # a z affinity on 2-D data becomes a scalar
    (["B", "C", "D"], 3, None, None),
]
