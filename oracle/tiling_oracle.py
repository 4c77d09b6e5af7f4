"""CPU oracle for the 3D overlap tiling path (crop -> spline window -> blend merge).

TEST INFRASTRUCTURE ONLY.  This file is a NumPy restatement of the reference algorithm and
is used solely as the checker by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``biapy_amd/`` imports it; the product
path is the HIP extension and fails loudly when that is missing.

Parity status: PINNED.  The reference holds no tests/golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
imported in the build container by ``tests/golden/make_golden.py`` and committed under
``tests/golden/tiling_*.npz`` (checked by ``tests/test_oracle_golden.py``).

Reference being restated (all paths relative to /root/reference):
  * grid arithmetic ............ biapy/data/data_3D_manipulation.py:536-563 (crop), :778-816 (merge)
  * crop ....................... biapy/data/data_3D_manipulation.py:505-533, :591-623
  * spline taper ............... biapy/data/data_3D_manipulation.py:662-688
  * blend / normalise .......... biapy/data/data_3D_manipulation.py:822-856
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


@dataclass(frozen=True)
class AxisGrid:
    """Patch placement along one axis (reference: data_3D_manipulation.py:541-547 / :789-795)."""

    n: int          # patches along the axis            (vols_per_*)
    step: int       # distance between patch starts     (step_* after redistribution)
    last: int       # shift applied to the final patch  (last_* after redistribution)
    patch: int      # patch extent used for the "does it still fit" test
    limit: int      # axis extent the test compares against
    ov_pixels: int  # (patch - 2*pad) - step, the taper length handed to the window

    def start(self, i: int) -> int:
        """Start of patch ``i`` (reference: d_* rule, :596-598 crop / :826-835 merge)."""
        d = 0 if (i * self.step + self.patch) < self.limit else self.last
        return i * self.step - d

    def starts(self) -> List[int]:
        return [self.start(i) for i in range(self.n)]


def axis_grid(dim: int, patch: int, pad: int, overlap: float, *, for_merge: bool) -> AxisGrid:
    """Grid along one axis.

    ``dim``   un-padded volume extent, ``patch`` the (padded) patch extent.
    For the crop the fit test runs in padded coordinates (patch vs dim+2*pad); for the merge it
    runs in original coordinates with the padding-stripped patch (patch-2*pad vs dim).
    """
    frac = 1 if overlap == 0 else 1 - overlap
    step = int((patch - pad * 2) * frac)
    n = math.ceil(dim / step)
    last = 0 if n == 1 else (((n - 1) * step) + patch) - (dim + 2 * pad)
    per_block = last // (n - 1) if n > 1 else 0
    step -= per_block
    last -= per_block * (n - 1)
    ov_pixels = (patch - pad * 2) - step
    if for_merge:
        return AxisGrid(n, step, last, patch - 2 * pad, dim, ov_pixels)
    return AxisGrid(n, step, last, patch, dim + 2 * pad, ov_pixels)


def check_overlap(overlap: Sequence[float]) -> None:
    if any((o >= 1 or o < 0) for o in overlap[:3]):
        raise ValueError("'overlap' values must be floats between range [0, 1)")


def crop_grid(vol_zyx: Sequence[int], patch_zyx: Sequence[int], overlap, padding) -> Tuple[AxisGrid, AxisGrid, AxisGrid]:
    return tuple(axis_grid(vol_zyx[a], patch_zyx[a], padding[a], overlap[a], for_merge=False) for a in range(3))


def merge_grid(vol_zyx: Sequence[int], patch_zyx: Sequence[int], overlap, padding) -> Tuple[AxisGrid, AxisGrid, AxisGrid]:
    return tuple(axis_grid(vol_zyx[a], patch_zyx[a], padding[a], overlap[a], for_merge=True) for a in range(3))


def crop_coords(vol_zyx, patch_zyx, overlap=(0, 0, 0), padding=(0, 0, 0)) -> np.ndarray:
    """(N,6) int64 rows ``z0,z1,y0,y1,x0,x1`` in the reference's patch order (z-major)."""
    check_overlap(overlap)
    gz, gy, gx = crop_grid(vol_zyx, patch_zyx, overlap, padding)
    out = np.empty((gz.n * gy.n * gx.n, 6), dtype=np.int64)
    c = 0
    for z in gz.starts():
        for y in gy.starts():
            for x in gx.starts():
                out[c] = (z, z + patch_zyx[0], y, y + patch_zyx[1], x, x + patch_zyx[2])
                c += 1
    return out


def pad_volume(data: np.ndarray, padding, pad_type: str = "reflect", median_padding: bool = False) -> np.ndarray:
    """np.pad step of the crop (reference :505-533), including the median overwrite quirk at :531
    (the y-far slab is indexed with ``data.shape[0]``, reproduced as is)."""
    mode = "constant" if pad_type == "zeros" else pad_type
    pz, py, px = padding
    out = np.pad(data, ((pz, pz), (py, py), (px, px), (0, 0)), mode)
    if median_padding:
        Z, Y, X = data.shape[:3]
        out[0:pz, :, :, :] = np.median(data[0, :, :, :])
        out[pz + Z : 2 * pz + Z, :, :, :] = np.median(data[-1, :, :, :])
        out[:, 0:py, :, :] = np.median(data[:, 0, :, :])
        out[:, py + Y : 2 * py + Z, :, :] = np.median(data[:, -1, :, :])
        out[:, :, 0:px, :] = np.median(data[:, :, 0, :])
        out[:, :, px + X : 2 * px + X, :] = np.median(data[:, :, -1, :])
    return out


def crop(data: np.ndarray, vol_shape, overlap=(0, 0, 0), padding=(0, 0, 0), pad_type="reflect", median_padding=False):
    """Returns (patches (N,Pz,Py,Px,C), coords (N,6))."""
    if data.ndim != 4:
        raise ValueError("data expected to be 4 dimensional, given {}".format(data.shape))
    coords = crop_coords(data.shape[:3], vol_shape[:3], overlap, padding)
    padded = pad_volume(data, padding, pad_type, median_padding)
    out = np.zeros((coords.shape[0],) + tuple(vol_shape[:3]) + (data.shape[-1],), dtype=data.dtype)
    for c, (z0, z1, y0, y1, x0, x1) in enumerate(coords):
        out[c] = padded[z0:z1, y0:y1, x0:x1]
    return out, coords


def taper_1d(size: int, ov_pixels: int, power: int = 2) -> np.ndarray:
    """1-D blend weights (reference :662-670): float64 rational taper stored into float32."""
    w = np.ones(size, dtype=np.float32)
    if ov_pixels > 0:
        ov = min(ov_pixels, size // 2)
        x = np.linspace(0, 1, ov + 2)[1:-1]
        t = (x ** power) / (x ** power + (1 - x) ** power + 1e-8)
        w[:ov] = t
        w[-ov:] = t[::-1]
    return w


def spline_window(patch_zyx, ov_pixels_zyx) -> np.ndarray:
    """(Pz,Py,Px,1) float32 window = fl32(fl32(wz*wy)*wx) (reference :673-688)."""
    wz = taper_1d(patch_zyx[0], ov_pixels_zyx[0])[:, None, None]
    wy = taper_1d(patch_zyx[1], ov_pixels_zyx[1])[None, :, None]
    wx = taper_1d(patch_zyx[2], ov_pixels_zyx[2])[None, None, :]
    return np.expand_dims(wz * wy * wx, -1).astype(np.float32)


def merge(data: np.ndarray, orig_vol_shape, data_mask: Optional[np.ndarray] = None, overlap=(0, 0, 0), padding=(0, 0, 0)):
    """Blend-merge (reference :754-856).  fp32 accumulation in z-major patch order."""
    assert data.ndim == 5
    assert len(orig_vol_shape) == 4
    if data_mask is not None and data.shape[:-1] != data_mask.shape[:-1]:
        raise ValueError("data and data_mask shapes mismatch: {} vs {}".format(data.shape[:-1], data_mask.shape[:-1]))
    check_overlap(overlap)
    full_patch = data.shape[1:4]
    pz, py, px = padding
    core = data[:, pz : data.shape[1] - pz, py : data.shape[2] - py, px : data.shape[3] - px, :]
    acc = np.zeros(tuple(orig_vol_shape), dtype=np.float32)
    acc_mask = None
    if data_mask is not None:
        core_mask = data_mask[:, pz : data_mask.shape[1] - pz, py : data_mask.shape[2] - py, px : data_mask.shape[3] - px, :]
        acc_mask = np.zeros(tuple(orig_vol_shape[:3]) + (core_mask.shape[-1],), dtype=np.float32)
    wsum = np.zeros(tuple(orig_vol_shape[:3]) + (1,), dtype=np.float32)
    gz, gy, gx = merge_grid(orig_vol_shape[:3], full_patch, overlap, padding)
    P = core.shape[1:4]
    win = spline_window(P, (gz.ov_pixels, gy.ov_pixels, gx.ov_pixels))
    c = 0
    for z0 in gz.starts():
        for y0 in gy.starts():
            for x0 in gx.starts():
                sl = (slice(z0, z0 + P[0]), slice(y0, y0 + P[1]), slice(x0, x0 + P[2]))
                acc[sl] += core[c] * win
                if acc_mask is not None:
                    acc_mask[sl] += core_mask[c] * win
                wsum[sl] += win
                c += 1
    merged = np.true_divide(acc, wsum + 1e-18).astype(data.dtype)
    if acc_mask is not None:
        return merged, np.true_divide(acc_mask, wsum + 1e-18).astype(data_mask.dtype)
    return merged


# ---------------------------------------------------------------------------------------------------
# 2D tiling (biapy/data/data_2D_manipulation.py:54-533)
# ---------------------------------------------------------------------------------------------------
# The 2D functions walk (image, y, x) with the SAME per-axis grid arithmetic (:194-211 crop, :466-483 merge), the same taper
# (:342-351) and the same fp32 blend (:497-516) as the 3D ones; an image stack (N,Y,X,C) is therefore a volume whose z axis
# has patch size 1, no overlap and no padding.  That embedding is pinned against the reference's own 2D outputs in
# tests/golden/tiling2d_golden.npz (tests/test_oracle_golden.py::test_tiling2d_*).
def crop2d(data: np.ndarray, crop_shape, overlap=(0, 0), padding=(0, 0), pad_type="reflect"):
    """Returns (patches (n,Py,Px,C), coords (n,4) = y0,y1,x0,x1) in the reference's (image, y, x) order."""
    if data.ndim != 4:
        raise ValueError("data expected to be 4 dimensional, given {}".format(data.shape))
    p, c = crop(data, (1, crop_shape[0], crop_shape[1], data.shape[-1]), (0.0, overlap[0], overlap[1]), (0, padding[0], padding[1]), pad_type)
    return p[:, 0], c[:, 2:]


def merge2d(data: np.ndarray, original_shape, data_mask: Optional[np.ndarray] = None, overlap=(0, 0), padding=(0, 0)):
    """(n,Py,Px,C) patches -> (N,Y,X,C) images (+ mask)."""
    dm = None if data_mask is None else data_mask[:, None]
    return merge(data[:, None], tuple(original_shape), dm, (0.0, overlap[0], overlap[1]), (0, padding[0], padding[1]))


# ---- the steps of process_test_sample around the blended prediction -----------------------------------------------------------------
def pad_to_shape(img: np.ndarray, crop_shape) -> np.ndarray:
    """biapy/data/data_manipulation.py:3218-3300 (mode "reflect"): every spatial axis shorter than ``crop_shape`` is extended IN FRONT with
    np.pad(..., "reflect"), one axis after the other (the image stays in the bottom-right corner).  Called by the test generators with
    DATA.REFLECT_TO_COMPLETE_SHAPE (generators/test_pair_data_generators.py:299-314)."""
    for ax in range(img.ndim - 1):
        if img.shape[ax] < crop_shape[ax]:
            pw = [(0, 0)] * img.ndim
            pw[ax] = (crop_shape[ax] - img.shape[ax], 0)
            img = np.pad(img, pw, "reflect")
    return img


def crop_to_reflected_orig_shape(pred: np.ndarray, orig_shape) -> np.ndarray:
    """biapy/engine/base_workflow.py:2089-2131 (3D branch): ``pred[-Z:, -Y:, -X:]`` - the prediction of the padded sample back to the
    sample's own extents (``pred`` here without the leading batch axis)."""
    z, y, x = orig_shape[:3]
    return pred[-z:, -y:, -x:]


def class_argmax(pred: np.ndarray, class_channels: int) -> np.ndarray:
    """biapy/engine/base_workflow.py:2135-2141: the trailing ``class_channels`` channels become one np.argmax channel (the concatenation
    with the float prediction makes it float)."""
    return np.concatenate((pred[..., :-class_channels], np.expand_dims(np.argmax(pred[..., -class_channels:], axis=-1), axis=-1)), axis=-1)
