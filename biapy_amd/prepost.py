"""Input normalisation and output binarisation on the device.

Mirrors ``biapy/data/norm.py`` (``percentile_clip`` :395-473, ``zero_mean_unit_variance_normalization`` :586-645) and the
binarisation of ``biapy/engine/semantic_seg.py:418-431`` (``threshold_otsu`` of the merged prediction, then ``pred > th``)
for float32 tensors that already live on the MI355X - the 1024^3 volume of cfg 3 is 4.3 GB, and the reference takes its
percentiles / histogram on the host.

* percentiles are EXACT: the two order statistics ``np.percentile`` interpolates between come from a 4-pass radix select
  (``bpx_select_kth_f32``) and the interpolation repeats NumPy's float32 ``_lerp`` arithmetic, so the bounds are bit-identical;
* the histogram repeats ``np.histogram``'s float32 bin arithmetic (``bpx_histogram_f32``): identical counts, hence the
  identical Otsu threshold;
* mean / std use double accumulation on the device (NumPy: float32 pairwise sums) - equal to ~1e-7 relative, not bitwise.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib as L

lib = L.lib


def _flat(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("biapy_amd.prepost runs on the MI355X only (tensor is on %s); there is no CPU path" % x.device)
    if x.dtype != torch.float32:
        raise NotImplementedError(f"biapy_amd.prepost works on float32 tensors (got {x.dtype})")
    return x.contiguous().view(-1)


def kth_values(x: torch.Tensor, ranks) -> list:
    """Exact order statistics x_sorted[k] (0-based) for every k in ``ranks`` as Python floats (one host sync at the end)."""
    f = _flat(x)
    ws = torch.empty(int(lib.bpx_select_workspace()), dtype=torch.uint8, device=f.device)
    out = torch.empty(len(ranks), dtype=torch.float32, device=f.device)
    for q, k in enumerate(ranks):
        L.check(lib.bpx_select_kth_f32(f.data_ptr(), f.numel(), int(k), out.data_ptr() + 4 * q, ws.data_ptr(), L.stream_ptr()))
    return [float(v) for v in out.cpu().numpy()]


def percentile(x: torch.Tensor, q: float) -> float:
    """``float(np.percentile(x, q))`` (method 'linear') of a float32 tensor without sorting: two exact order statistics, then
    NumPy's arithmetic repeated operation by operation.  For a float32 array NumPy (>= 2) works in float32 throughout:
    ``quantile = q / float32(100)``, virtual index ``(n - 1) * quantile``, ``gamma = index - floor(index)`` and the ``_lerp``
    ``a + (b - a) * gamma`` (``b - (b - a) * (1 - gamma)`` for gamma >= 0.5) - numpy/lib/_function_base_impl.py."""
    n = x.numel()
    vi = np.float32(n - 1) * (np.float32(q) / np.float32(100))
    lo = int(math.floor(float(vi)))
    hi = min(lo + 1, n - 1)
    t = np.float32(vi - np.float32(lo))
    a, b = (np.float32(v) for v in kth_values(x, [lo, hi]))
    d = np.float32(b - a)
    if t >= 0.5:
        return float(np.float32(b - np.float32(d * np.float32(np.float32(1) - t))))
    return float(np.float32(a + np.float32(d * t)))


def minmax(x: torch.Tensor) -> Tuple[float, float]:
    f = _flat(x)
    part = torch.empty((lib.bpx_scan_blocks(f.numel()), 2), dtype=torch.float32, device=f.device)
    L.check(lib.bpx_minmax_f32(f.data_ptr(), f.numel(), part.data_ptr(), L.stream_ptr()))
    return float(part[:, 0].min()), float(part[:, 1].max())


def _is_binary(x: torch.Tensor) -> bool:
    mn, mx = minmax(x)      # norm.py:38-42: a channel holding only 0 / 1 is never clipped or normalised
    return mn >= 0.0 and mx <= 1.0 and bool(((x == 0) | (x == 1)).all())


def percentile_clip(data: torch.Tensor, per_lower_bound: Optional[float] = None, per_upper_bound: Optional[float] = None,
                    lower_bound_val: Optional[float] = None, upper_bound_val: Optional[float] = None, apply_norm: bool = True):
    """Device counterpart of ``biapy.data.norm.percentile_clip`` on its NumPy branch (bounds from ``np.percentile``)."""
    if _is_binary(data):
        return data, 0.0, 1.0
    if per_lower_bound is None or per_lower_bound == -1:
        assert lower_bound_val is not None, "If 'per_lower_bound' is not provided, 'lower_bound_val' should be provided"
        x_lwr = lower_bound_val
    else:
        assert per_lower_bound > 0, "Value in 'per_lower_bound' should be less than 100"
        x_lwr = percentile(data, per_lower_bound)
    if per_upper_bound is None or per_upper_bound == -1:
        assert upper_bound_val is not None, "If 'per_upper_bound' is not provided, 'upper_bound_val' should be provided"
        x_upr = upper_bound_val
    else:
        assert per_upper_bound < 100, "Value in 'per_upper_bound' should be less than 100"
        x_upr = percentile(data, per_upper_bound)
    if apply_norm:
        out = torch.empty_like(data, memory_format=torch.contiguous_format)
        L.check(lib.bpx_clip_affine_f32(_flat(data).data_ptr(), data.numel(), x_lwr, x_upr, 0.0, 1.0, out.data_ptr(), L.stream_ptr()))
        data = out
    return data, x_lwr, x_upr


def mean_std(x: torch.Tensor) -> Tuple[float, float]:
    """Population mean / std (``ndarray.mean()`` / ``ndarray.std()``), two passes, double accumulation."""
    f = _flat(x)
    nb = lib.bpx_scan_blocks(f.numel())
    part = torch.empty(nb, dtype=torch.float64, device=f.device)
    L.check(lib.bpx_moment_f32(f.data_ptr(), f.numel(), 0.0, 1, part.data_ptr(), L.stream_ptr()))
    mean = float(part.sum()) / f.numel()
    L.check(lib.bpx_moment_f32(f.data_ptr(), f.numel(), mean, 2, part.data_ptr(), L.stream_ptr()))
    return mean, math.sqrt(float(part.sum()) / f.numel())


def zero_mean_unit_variance_normalization(data: torch.Tensor, mean: Optional[float] = None, std: Optional[float] = None,
                                          apply_norm: bool = True, eps: float = 1e-6):
    """Device counterpart of ``biapy.data.norm.zero_mean_unit_variance_normalization`` (NumPy branch)."""
    assert data.dim() >= 2, "Data should be at least 2D. E.g. (y, x) in 2D and (z, y, x) in 3D"
    if _is_binary(data):
        return data, 0.0, 1.0
    if mean is None or std is None:
        m, s = mean_std(data)
        mean = m if mean is None else mean
        std = s if std is None else std
    if apply_norm:
        out = torch.empty_like(data, memory_format=torch.contiguous_format)
        L.check(lib.bpx_clip_affine_f32(_flat(data).data_ptr(), data.numel(), -math.inf, math.inf, mean, max(std, eps), out.data_ptr(),
                                        L.stream_ptr()))
        data = out
    return data, float(mean), float(std)


def histogram(x: torch.Tensor, nbins: int = 256) -> Tuple[np.ndarray, np.ndarray]:
    """``np.histogram(x, bins=nbins, range=(x.min(), x.max()))`` of a float32 tensor: (counts int64, float32 edges)."""
    f = _flat(x)
    mn, mx = minmax(x)
    first, last = np.float32(mn), np.float32(mx)
    if first == last:                                    # numpy widens an empty range by +-0.5
        first, last = np.float32(first - np.float32(0.5)), np.float32(last + np.float32(0.5))
    edges = np.linspace(first, last, nbins + 1, endpoint=True, dtype=np.float32)
    ed = torch.from_numpy(edges).to(f.device)
    counts = torch.zeros(nbins, dtype=torch.int64, device=f.device)
    L.check(lib.bpx_histogram_f32(f.data_ptr(), f.numel(), float(first), float(last), nbins, ed.data_ptr(), counts.data_ptr(), L.stream_ptr()))
    return counts.cpu().numpy(), edges


def threshold_otsu(x: torch.Tensor, nbins: int = 256) -> float:
    """skimage.filters.threshold_otsu on the device histogram (Otsu 1979: arg-max of the between-class variance over the
    bin centres).  The counts are bit-identical to np.histogram's, the few float64 operations below follow scikit-image."""
    counts, edges = histogram(x, nbins)
    if float(edges[0]) == float(edges[-1]):
        return float(edges[0])
    centers = (edges[:-1] + edges[1:]) / 2
    c = counts.astype(np.float64)
    w1, w2 = np.cumsum(c), np.cumsum(c[::-1])[::-1]
    m1 = np.cumsum(c * centers) / w1
    m2 = (np.cumsum((c * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return float(centers[int(np.argmax(var12))])


def binarize(pred: torch.Tensor, threshold: Optional[float] = None) -> torch.Tensor:
    """``(pred > threshold_otsu(pred)).astype(np.uint8)`` (semantic_seg.py:425-427) as a uint8 device tensor."""
    th = threshold_otsu(pred) if threshold is None else float(threshold)
    f = _flat(pred)
    out = torch.empty(pred.shape, dtype=torch.uint8, device=pred.device)
    L.check(lib.bpx_threshold_u8(f.data_ptr(), f.numel(), th, out.data_ptr(), L.stream_ptr()))
    return out
