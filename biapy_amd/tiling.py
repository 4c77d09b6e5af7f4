"""Drop-ins for BiaPy's 3D overlap tiling, running on the MI355X.

``crop_3D_data_with_overlap`` / ``merge_3D_data_with_overlap`` keep the reference's signatures, return
arities, error conditions and (bit-exact) results - biapy/data/data_3D_manipulation.py:353-636 and
:690-859 - but the copy / blend loops run as HIP kernels (bpx_crop3d_gather, bpx_merge3d_blend).
NumPy in -> NumPy out as in the reference; the ``*_device`` functions underneath work on tensors that
are already resident in HBM and are what the sliding-window engine uses (no host round trip).

Only the integer grid arithmetic and the 1-D taper vectors are computed on the host (a few hundred
scalars); everything that touches voxel data is on the device.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L

__all__ = [
    "PatchCoords", "crop_3D_data_with_overlap", "merge_3D_data_with_overlap", "axis_grid", "crop_grid", "merge_grid",
    "taper_1d", "crop_device", "crop_rows_needed", "merge_device",
]


class PatchCoords:
    """Same attribute contract as biapy/data/dataset.py:484-544."""

    def __init__(self, y_start, y_end, x_start, x_end, z_start=None, z_end=None):
        self.y_start, self.y_end, self.x_start, self.x_end = y_start, y_end, x_start, x_end
        if z_start is not None:
            self.z_start = z_start
        if z_end is not None:
            self.z_end = z_end

    def extract_shape_from_coords(self):
        shape = []
        if hasattr(self, "z_start") and hasattr(self, "z_end"):
            shape += [self.z_end - self.z_start]
        return tuple(shape + [self.y_end - self.y_start, self.x_end - self.x_start])

    def __repr__(self):
        z = f"{self.z_start}:{self.z_end}," if hasattr(self, "z_start") else ""
        return f"[{z}{self.y_start}:{self.y_end},{self.x_start}:{self.x_end}]"


# ---------------------------------------------------------------------------------------------------
# host-side integer geometry (reference: data_3D_manipulation.py:536-563 and :778-816)
# ---------------------------------------------------------------------------------------------------
def axis_grid(dim: int, patch: int, pad: int, overlap: float, for_merge: bool) -> L.AxisGrid:
    frac = 1 if overlap == 0 else 1 - overlap
    step = int((patch - pad * 2) * frac)
    n = math.ceil(dim / step)
    last = 0 if n == 1 else (((n - 1) * step) + patch) - (dim + 2 * pad)
    per_block = last // (n - 1) if n > 1 else 0
    step -= per_block
    last -= per_block * (n - 1)
    if for_merge:
        return L.AxisGrid(n, step, last, patch - 2 * pad, dim)
    return L.AxisGrid(n, step, last, patch, dim + 2 * pad)


def _start(g: L.AxisGrid, i: int) -> int:
    s = i * g.step
    return s - (0 if s + g.patch < g.limit else g.last)


def crop_grid(vol_zyx, patch_zyx, overlap, padding):
    return (L.AxisGrid * 3)(*[axis_grid(vol_zyx[a], patch_zyx[a], padding[a], overlap[a], False) for a in range(3)])


def merge_grid(vol_zyx, patch_zyx, overlap, padding):
    return (L.AxisGrid * 3)(*[axis_grid(vol_zyx[a], patch_zyx[a], padding[a], overlap[a], True) for a in range(3)])


def taper_1d(size: int, ov_pixels: int, power: int = 2) -> np.ndarray:
    """float64 rational taper stored to float32 (reference :662-670) - 1-D, a few hundred scalars."""
    w = np.ones(size, dtype=np.float32)
    if ov_pixels > 0:
        ov = min(ov_pixels, size // 2)
        x = np.linspace(0, 1, ov + 2)[1:-1]
        t = (x ** power) / (x ** power + (1 - x) ** power + 1e-8)
        w[:ov] = t
        w[-ov:] = t[::-1]
    return w


def _check_overlap(overlap):
    if (overlap[0] >= 1 or overlap[0] < 0) or (overlap[1] >= 1 or overlap[1] < 0) or (overlap[2] >= 1 or overlap[2] < 0):
        raise ValueError("'overlap' values must be floats between range [0, 1)")


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("biapy_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


# ---------------------------------------------------------------------------------------------------
# device-level API
# ---------------------------------------------------------------------------------------------------
def crop_rows_needed(vol_zyx, patch_zyx, overlap, padding, zrow_lo: int, zrow_hi: int, reflect: bool = True):
    """Input slices [z_lo, z_hi) of the UN-padded volume that the patches of z-rows [zrow_lo, zrow_hi) read, including the
    sources of the reflect padding at the two ends of the volume (np.pad "reflect", data_3D_manipulation.py:505-515)."""
    Z, pz, Pz = int(vol_zyx[0]), int(padding[0]), int(patch_zyx[0])
    g = axis_grid(Z, Pz, pz, overlap[0], False)
    # the hull over EVERY row: the reference's shift-back of the last rows can make the starts non-monotonic (0, 2, 4, 6, 4, 6, 8)
    starts = [_start(g, r) for r in range(zrow_lo, zrow_hi)]
    a = min(starts) - pz
    b = max(starts) - pz + Pz
    lo, hi = max(a, 0), min(b, Z)
    if reflect and a < 0:
        hi = max(hi, min(Z, -a + 1))
    if reflect and b > Z:
        lo = min(lo, max(0, 2 * (Z - 1) - (b - 1)))
    return lo, hi


def crop_device(vol: torch.Tensor, patch_zyx: Sequence[int], overlap=(0, 0, 0), padding=(0, 0, 0), pad_type: str = "reflect",
                c_begin: int = 0, c_count: Optional[int] = None, out: Optional[torch.Tensor] = None, z_offset: int = 0,
                full_z: Optional[int] = None) -> torch.Tensor:
    """vol: (Z,Y,X,C) device tensor (1/2/4-byte dtype) -> (n,Pz,Py,Px,C) patches [c_begin, c_begin+c_count).

    Slab form (sharded sliding window: a rank holds only the input slices its patches read): ``vol`` holds the slices
    [z_offset, z_offset + vol.shape[0]) of a volume with ``full_z`` slices; the patch grid is that of the full volume and the
    requested patches must only read slices of the slab (``crop_rows_needed``) - checked here, not in the kernel."""
    assert vol.is_cuda and vol.dim() == 4 and vol.is_contiguous()
    Zs, Y, X, Cc = vol.shape
    Z = Zs if full_z is None else int(full_z)
    g = crop_grid((Z, Y, X), patch_zyx, overlap, padding)
    n_all = g[0].n * g[1].n * g[2].n
    if c_count is None:
        c_count = n_all - c_begin
    if out is None:
        out = torch.empty((c_count, patch_zyx[0], patch_zyx[1], patch_zyx[2], Cc), dtype=vol.dtype, device=vol.device)
    mode = 1 if pad_type == "zeros" else 0
    if pad_type not in ("reflect", "zeros"):
        raise ValueError(f"pad_type {pad_type!r} is not supported on the device path (reflect|zeros)")
    base = vol.data_ptr()
    if z_offset or Z != Zs:
        if c_count > 0:
            per_row = g[1].n * g[2].n
            lo, hi = crop_rows_needed((Z, Y, X), patch_zyx, overlap, padding, c_begin // per_row, (c_begin + c_count - 1) // per_row + 1,
                                      reflect=(mode == 0))
            if lo < z_offset or hi > z_offset + Zs:
                raise ValueError(f"patches [{c_begin}, {c_begin + c_count}) read slices [{lo}, {hi}) but the slab holds "
                                 f"[{z_offset}, {z_offset + Zs})")
        base -= z_offset * Y * X * Cc * vol.element_size()             # virtual origin of the full volume: only slab rows are dereferenced
    L.check(L.lib.bpx_crop3d_gather(base, vol.element_size(), Z, Y, X, Cc, padding[0], padding[1], padding[2], mode, g,
                                    c_begin, c_count, out.data_ptr(), L.stream_ptr()))
    return out


class MergePlan:
    """Host-side constants of one merge geometry: grid + the three taper vectors on the device."""

    def __init__(self, vol_zyx, full_patch_zyx, overlap, padding, device):
        self.vol = tuple(int(v) for v in vol_zyx)
        self.full_patch = tuple(int(p) for p in full_patch_zyx)
        self.padding = tuple(int(p) for p in padding)
        self.grid = merge_grid(self.vol, self.full_patch, overlap, self.padding)
        core = [self.full_patch[a] - 2 * self.padding[a] for a in range(3)]
        ovpx = [core[a] - self.grid[a].step for a in range(3)]
        self.w = [torch.from_numpy(taper_1d(core[a], ovpx[a])).to(device) for a in range(3)]
        self.n_patches = self.grid[0].n * self.grid[1].n * self.grid[2].n

    def z_rows(self):
        return self.grid[0].n

    def row_start(self, iz):
        return _start(self.grid[0], iz)


def merge_device(patches: torch.Tensor, plan: MergePlan, out_dtype: Optional[torch.dtype] = None, z_lo: int = 0,
                 z_hi: Optional[int] = None, zrow_lo: int = 0, zrow_hi: Optional[int] = None, acc: Optional[torch.Tensor] = None,
                 wacc: Optional[torch.Tensor] = None, seed: bool = False, write_partial: bool = False,
                 out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """patches: (n,Pz,Py,Px,C) on the device holding patch rows [zrow_lo, zrow_hi) -> (z_hi-z_lo, Y, X, C)."""
    assert patches.is_cuda and patches.dim() == 5 and patches.is_contiguous()
    Z, Y, X = plan.vol
    z_hi = Z if z_hi is None else z_hi
    zrow_hi = plan.grid[0].n if zrow_hi is None else zrow_hi
    Cc = patches.shape[-1]
    assert tuple(patches.shape[1:4]) == plan.full_patch
    assert patches.shape[0] == (zrow_hi - zrow_lo) * plan.grid[1].n * plan.grid[2].n, "patch count does not match the row range"
    out_dtype = out_dtype or patches.dtype
    flags = (1 if write_partial else 0) | (2 if seed else 0)
    if not write_partial and out is None:
        out = torch.empty((z_hi - z_lo, Y, X, Cc), dtype=out_dtype, device=patches.device)
    L.check(L.lib.bpx_merge3d_blend(
        patches.data_ptr(), L.dt_of(patches), plan.full_patch[0], plan.full_patch[1], plan.full_patch[2], Cc,
        plan.padding[0], plan.padding[1], plan.padding[2], plan.grid, plan.w[0].data_ptr(), plan.w[1].data_ptr(), plan.w[2].data_ptr(),
        Z, Y, X, z_lo, z_hi, zrow_lo, zrow_hi, L.ptr(acc), L.ptr(wacc), flags, L.ptr(out), L.dt_of(out) if out is not None else 0,
        L.stream_ptr()))
    return out


# ---------------------------------------------------------------------------------------------------
# NumPy drop-ins with the reference's signatures
# ---------------------------------------------------------------------------------------------------
def _face_median(t: torch.Tensor):
    # np.median semantics on the device: mean of the two middle order statistics (np.mean of two
    # float32 values = fl32(fl32(a+b)/2)); integer inputs are not supported by this option.
    s, _ = torch.sort(t.flatten())
    n = s.numel()
    if n % 2:
        return s[n // 2]
    return (s[n // 2 - 1] + s[n // 2]) / 2


def crop_3D_data_with_overlap(data, vol_shape, data_mask=None, overlap=(0, 0, 0), padding=(0, 0, 0), verbose=True,
                              median_padding=False, load_data=True, pad_type="reflect"):
    """Drop-in for biapy.data.data_3D_manipulation.crop_3D_data_with_overlap (same checks, same returns)."""
    if verbose:
        print("### 3D-OV-CROP ###")
        print("Cropping {} images into {} with overlapping . . .".format(data.shape, vol_shape))
        print("Minimum overlap selected: {}".format(overlap))
        print("Padding: {}".format(padding))
    if data.ndim != 4:
        raise ValueError("data expected to be 4 dimensional, given {}".format(data.shape))
    if data_mask is not None:
        if data_mask.ndim != 4:
            raise ValueError("data_mask expected to be 4 dimensional, given {}".format(data_mask.shape))
        if data.shape[:-1] != data_mask.shape[:-1]:
            raise ValueError("data and data_mask shapes mismatch: {} vs {}".format(data.shape[:-1], data_mask.shape[:-1]))
    if len(vol_shape) != 4:
        raise ValueError("vol_shape expected to be of length 4, given {}".format(vol_shape))
    for i, p in enumerate(padding):
        if p >= vol_shape[i] // 2:
            raise ValueError(
                "'Padding' can not be greater than half of 'vol_shape'. Max value for the given input shape {} is {}".format(
                    vol_shape, ((vol_shape[0] // 2) - 1, (vol_shape[1] // 2) - 1, (vol_shape[2] // 2) - 1)))
    for a in range(3):
        if vol_shape[a] > data.shape[a]:
            raise ValueError(
                "'vol_shape[{}]' {} greater than {} (you can reduce 'DATA.PATCH_SIZE' or use 'DATA.REFLECT_TO_COMPLETE_SHAPE')".format(
                    a, vol_shape[a], data.shape[a]))
    _check_overlap(overlap)

    g = crop_grid(data.shape[:3], vol_shape[:3], overlap, padding)
    if verbose:
        core = [vol_shape[a] - 2 * padding[a] for a in range(3)]
        print("{} patches per (z,y,x) axis".format((g[0].n, g[1].n, g[2].n)))
        del core
    coords: List[PatchCoords] = []
    for iz in range(g[0].n):
        z0 = _start(g[0], iz)
        for iy in range(g[1].n):
            y0 = _start(g[1], iy)
            for ix in range(g[2].n):
                x0 = _start(g[2], ix)
                coords.append(PatchCoords(z_start=z0, z_end=z0 + vol_shape[0], y_start=y0, y_end=y0 + vol_shape[1],
                                          x_start=x0, x_end=x0 + vol_shape[2]))
    if not load_data:
        if verbose:
            print("### END 3D-OV-CROP ###")
        return coords

    dev = _device()

    def run(arr: np.ndarray, med: bool) -> np.ndarray:
        if arr.dtype.itemsize not in (1, 2, 4):
            raise ValueError(f"dtype {arr.dtype} is not supported by the device crop (1/2/4-byte element types only)")
        if pad_type not in ("reflect", "zeros"):
            raise ValueError(f"pad_type {pad_type!r} is not supported on the device path (reflect|zeros)")
        if med:
            # median padding overwrites whole faces of the padded volume (reference :527-533, incl. the
            # `data.shape[0]` index quirk at :531).  All on the device: the padded volume is one
            # "patch" of the gather kernel, the six faces are overwritten with device-side medians,
            # and the patches are then gathered from it with no further padding.
            Z, Y, X = arr.shape[:3]
            pz, py, px = padding
            tv = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
            whole = (L.AxisGrid * 3)(L.AxisGrid(1, 1, 0, Z + 2 * pz, Z + 2 * pz), L.AxisGrid(1, 1, 0, Y + 2 * py, Y + 2 * py),
                                     L.AxisGrid(1, 1, 0, X + 2 * px, X + 2 * px))
            padded = torch.empty((Z + 2 * pz, Y + 2 * py, X + 2 * px, arr.shape[-1]), dtype=tv.dtype, device=dev)
            L.check(L.lib.bpx_crop3d_gather(tv.data_ptr(), tv.element_size(), Z, Y, X, arr.shape[-1], pz, py, px,
                                            1 if pad_type == "zeros" else 0, whole, 0, 1, padded.data_ptr(), L.stream_ptr()))
            padded[0:pz, :, :, :] = _face_median(tv[0])
            padded[pz + Z: 2 * pz + Z, :, :, :] = _face_median(tv[-1])
            padded[:, 0:py, :, :] = _face_median(tv[:, 0])
            padded[:, py + Y: 2 * py + Z, :, :] = _face_median(tv[:, -1])
            padded[:, :, 0:px, :] = _face_median(tv[:, :, 0])
            padded[:, :, px + X: 2 * px + X, :] = _face_median(tv[:, :, -1])
            out = torch.empty((len(coords),) + tuple(vol_shape[:3]) + (arr.shape[-1],), dtype=tv.dtype, device=dev)
            L.check(L.lib.bpx_crop3d_gather(padded.data_ptr(), padded.element_size(), padded.shape[0], padded.shape[1], padded.shape[2],
                                            padded.shape[3], 0, 0, 0, 1, g, 0, len(coords), out.data_ptr(), L.stream_ptr()))
            return out.cpu().numpy()
        t = torch.from_numpy(np.ascontiguousarray(arr).view(_uint_view(arr.dtype))).to(dev)
        out = crop_device(t, vol_shape[:3], overlap, padding, pad_type)
        return out.cpu().numpy().view(arr.dtype)

    cropped = run(data, median_padding)
    cropped_mask = run(data_mask, False) if data_mask is not None else None
    if verbose:
        print("**** New data shape is: {}".format(cropped.shape))
        print("### END 3D-OV-CROP ###")
    if data_mask is not None:
        return cropped, cropped_mask, coords
    return cropped, coords


def _uint_view(dt: np.dtype):
    return {1: np.uint8, 2: np.int16, 4: np.int32}[np.dtype(dt).itemsize]


_MERGE_DT = {np.dtype(np.float32): torch.float32, np.dtype(np.float16): torch.float16, np.dtype(np.uint8): torch.uint8}


def merge_3D_data_with_overlap(data, orig_vol_shape, data_mask=None, overlap=(0, 0, 0), padding=(0, 0, 0), verbose=True):
    """Drop-in for biapy.data.data_3D_manipulation.merge_3D_data_with_overlap (bit-exact, deterministic)."""
    assert data.ndim == 5, f"data expected to be 5 dimensional, given {data.shape}"
    assert len(orig_vol_shape) == 4, f"orig_vol_shape expected to be 4 dimensional, given {orig_vol_shape}"
    if data_mask is not None:
        if data.shape[:-1] != data_mask.shape[:-1]:
            raise ValueError("data and data_mask shapes mismatch: {} vs {}".format(data.shape[:-1], data_mask.shape[:-1]))
    _check_overlap(overlap)
    if verbose:
        print("### MERGE-3D-OV-CROP ###")
        print("Merging {} images into {} with smooth blending . . .".format(data.shape, orig_vol_shape))
        print("Minimum overlap selected: {}".format(overlap))
        print("Padding: {}".format(padding))
    dev = _device()
    plan = MergePlan(orig_vol_shape[:3], data.shape[1:4], overlap, padding, dev)
    if plan.n_patches != data.shape[0]:
        raise ValueError(f"expected {plan.n_patches} patches for volume {tuple(orig_vol_shape)}, got {data.shape[0]}")

    def run(arr: np.ndarray) -> np.ndarray:
        if arr.dtype not in _MERGE_DT:
            raise ValueError(f"dtype {arr.dtype} is not supported by the device merge (float32, float16, uint8)")
        if arr.shape[-1] != (orig_vol_shape[3] if arr is data else arr.shape[-1]):
            raise ValueError("channel count of data does not match orig_vol_shape")
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        return merge_device(t, plan).cpu().numpy()

    merged = run(data)
    if verbose:
        print("**** New data shape is: {}".format(merged.shape))
        print("### END MERGE-3D-OV-CROP ###")
    if data_mask is not None:
        return merged, run(data_mask)
    return merged


# ---------------------------------------------------------------------------------------------------
# 2D drop-ins (biapy/data/data_2D_manipulation.py:54-316 crop_data_with_overlap, :366-533 merge_data_with_overlap)
# ---------------------------------------------------------------------------------------------------
# An image stack (N,Y,X,C) is a volume whose z axis has patch size 1, no overlap and no padding: the reference's 2D code
# uses the same per-axis grid arithmetic, taper and fp32 blend as its 3D code, so the 3D gather / blend kernels serve both
# (bit-exact against the reference's 2D outputs: tests/golden/tiling2d_golden.npz).
def crop_data_with_overlap(data, crop_shape, data_mask=None, overlap=(0, 0), padding=(0, 0), verbose=True, load_data=True,
                           pad_type="reflect"):
    """Drop-in for biapy.data.data_2D_manipulation.crop_data_with_overlap (same checks, messages, returns)."""
    if data.ndim != 4:
        raise ValueError("data expected to be 4 dimensional, given {}".format(data.shape))
    if data_mask is not None:
        if data.ndim != 4:
            raise ValueError("data mask expected to be 4 dimensional, given {}".format(data_mask.shape))
        if data.shape[:-1] != data_mask.shape[:-1]:
            raise ValueError("data and data_mask shapes mismatch: {} vs {}".format(data.shape[:-1], data_mask.shape[:-1]))
    for i, p in enumerate(padding):
        if p >= crop_shape[i] // 2:
            raise ValueError("'Padding' can not be greater than the half of 'crop_shape'. Max value for this {} input shape is {}".format(
                crop_shape, ((crop_shape[0] // 2) - 1, (crop_shape[1] // 2) - 1)))
    if len(crop_shape) != 3:
        raise ValueError("crop_shape expected to be of length 3, given {}".format(crop_shape))
    for a in range(2):
        if crop_shape[a] > data.shape[a + 1]:
            raise ValueError("'crop_shape[{}]' {} greater than {} (you can reduce 'DATA.PATCH_SIZE' or use 'DATA.REFLECT_TO_COMPLETE_SHAPE')".format(
                a, crop_shape[a], data.shape[a + 1]))
    if (overlap[0] >= 1 or overlap[0] < 0) or (overlap[1] >= 1 or overlap[1] < 0):
        raise ValueError("'overlap' values must be floats between range [0, 1)")
    if verbose:
        print("### OV-CROP ###")
        print("Cropping {} images into {} with overlapping. . .".format(data.shape, crop_shape))
        print("Minimum overlap selected: {}".format(overlap))
        print("Padding: {}".format(padding))
    patch3, ov3, pad3 = (1, crop_shape[0], crop_shape[1]), (0.0, overlap[0], overlap[1]), (0, padding[0], padding[1])
    g = crop_grid(data.shape[:3], patch3, ov3, pad3)
    if verbose:
        print("{} patches per (y,x) axis".format((g[1].n, g[2].n)))
    coords: List[PatchCoords] = []
    for _ in range(g[0].n):
        for iy in range(g[1].n):
            y0 = _start(g[1], iy)
            for ix in range(g[2].n):
                x0 = _start(g[2], ix)
                coords.append(PatchCoords(y_start=y0, y_end=y0 + crop_shape[0], x_start=x0, x_end=x0 + crop_shape[1]))
    if not load_data:
        return coords
    dev = _device()

    def run(arr: np.ndarray) -> np.ndarray:
        if arr.dtype.itemsize not in (1, 2, 4):
            raise ValueError(f"dtype {arr.dtype} is not supported by the device crop (1/2/4-byte element types only)")
        t = torch.from_numpy(np.ascontiguousarray(arr).view(_uint_view(arr.dtype))).to(dev)
        out = crop_device(t, patch3, ov3, pad3, pad_type)
        return out.cpu().numpy().view(arr.dtype)[:, 0]

    cropped = run(data)
    cropped_mask = run(data_mask) if data_mask is not None else None
    if verbose:
        print("**** New data shape is: {}".format(cropped.shape))
        print("### END OV-CROP ###")
    if data_mask is not None:
        return cropped, cropped_mask, coords
    return cropped, coords


def merge_data_with_overlap(data, original_shape, data_mask=None, overlap=(0, 0), padding=(0, 0), verbose=True):
    """Drop-in for biapy.data.data_2D_manipulation.merge_data_with_overlap (bit-exact, deterministic)."""
    if data_mask is not None:
        if data.shape[:-1] != data_mask.shape[:-1]:
            raise ValueError("data and data_mask shapes mismatch: {} vs {}".format(data.shape[:-1], data_mask.shape[:-1]))
    for i, p in enumerate(padding):
        if p >= data.shape[i + 1] // 2:
            raise ValueError(f"'Padding' cannot be greater than half of 'data' shape. Max value for this {data.shape} input shape is "
                             f"{(data.shape[1] // 2) - 1, (data.shape[2] // 2) - 1}")
    if (overlap[0] >= 1 or overlap[0] < 0) or (overlap[1] >= 1 or overlap[1] < 0):
        raise ValueError("'overlap' values must be floats between range [0, 1)")
    if verbose:
        print("### MERGE-OV-CROP ###")
        print(f"Merging {data.shape} images into {original_shape} with smooth blending . . .")
        print(f"Overlap selected: {overlap}")
        print(f"Padding: {padding}")
    dev = _device()
    plan = MergePlan(original_shape[:3], (1, data.shape[1], data.shape[2]), (0.0, overlap[0], overlap[1]), (0, padding[0], padding[1]), dev)
    if plan.n_patches != data.shape[0]:
        raise ValueError(f"expected {plan.n_patches} patches for images {tuple(original_shape)}, got {data.shape[0]}")

    def run(arr: np.ndarray) -> np.ndarray:
        if arr.dtype not in _MERGE_DT:
            raise ValueError(f"dtype {arr.dtype} is not supported by the device merge (float32, float16, uint8)")
        t = torch.from_numpy(np.ascontiguousarray(arr[:, None])).to(dev)
        return merge_device(t, plan).cpu().numpy()

    merged = run(data)
    if verbose:
        print(f"**** New data shape is: {merged.shape}")
        print("### END MERGE-OV-CROP ###")
    if data_mask is not None:
        return merged, run(data_mask)
    return merged
