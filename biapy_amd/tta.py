"""Test-time augmentation on the device.

Mirrors ``biapy/data/post_processing/tta.py`` (``AxisTransform`` :64-196, ``build_axis_transform_group`` :197-256) and the
scalar-field case of ``ensemble_predictions`` (``biapy/data/post_processing/post_processing.py:1386-1540``; what semantic
segmentation uses - ``tta_spec=None``): predict the patch in every orientation (8 in 2D, 16 in 3D: flips x in-plane 90 degree
rotations, Z is never swapped), undo each orientation, reduce with mean / min / max.

The volume stays on the MI355X: one gather per orientation (``bpx_tta_orient``) and one fused un-orient + reduce pass
(``bpx_tta_accumulate``) instead of the reference's NumPy stack of 16 copies; the reduction runs in the reference's orientation
order, so the float32 mean is the same sequential sum ``np.mean(stack, axis=0)`` forms.
Direction-carrying channels (round 3; ``tta_spec``, tta.py:270-640): flows / offsets (``VectorChannels`` signed), per-axis magnitudes
(unsigned), StarDist rays (``RayChannels``) and affinities (``AffinityChannels``) - the spec drops the orientations a representation cannot
express, the volume is padded with zeros instead of reflections, every un-oriented prediction gets the group's channel remap (a signed
permutation of components, a permutation of ray channels, a permutation + roll of affinity maps) before the reduction, and min / max
apply to the mode-reducible channels only.  ``tta_spec`` may be the reference's ``TTASpec`` object or the plain dataclasses below (the
same field names); pinned to the reference's classes driven through its ``ensemble_predictions`` (tests/golden/tta_spec_golden.npz).
"""
from __future__ import annotations

import ctypes as C
import itertools
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

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


# ---- channel groups of a TTA spec (host-side descriptions; field names of biapy/data/post_processing/tta.py:318-540) -------------------
@dataclass
class ScalarChannels:
    channels: Tuple[int, ...] = ()
    name: str = "scalar"


@dataclass
class VectorChannels:
    axis_channels: Tuple[Optional[int], ...] = ()
    signed: bool = True
    axis_scale: Optional[Tuple[float, ...]] = None
    name: str = "vector"


@dataclass
class RayChannels:
    start: int = 0
    dirs: "np.ndarray" = field(default_factory=lambda: np.zeros((0, 2), np.float32))
    name: str = "rays"


@dataclass
class AffinityChannels:
    layout: Dict[Tuple[int, int], int] = field(default_factory=dict)
    name: str = "affinities"


@dataclass
class TTASpec:
    ndim: int
    n_channels: int
    groups: List[object] = field(default_factory=list)


# ---- the spec from the model's output channel names (tta.py:642-866) ------------------------------------------------------------------
_AXIS_LETTERS = {2: ("y", "x"), 3: ("z", "y", "x")}
# single-channel vector components: name -> (family, axis letter).  Flows of Cellpose / Omnipose and the HoVer maps.
_COMPONENTS = {"Gz": ("flow", "z"), "Gv": ("flow", "y"), "Gh": ("flow", "x"), "Z": ("hover", "z"), "V": ("hover", "y"), "H": ("hover", "x")}


def parse_model_output_channel_names(model_output_channel_info: Sequence[str]) -> List[str]:
    """One name per physical channel of ``pred`` from the per-head "+"-joined descriptions; the separate "class" head is not part of ``pred``
    (tta.py:674-698)."""
    return [c for head in model_output_channel_info if head != "class" for c in head.split("+") if c]


def _generate_rays(n_rays: int, ndim: int) -> "np.ndarray":
    """Unit ray directions in Cartesian (x, y[, z]) order, float32: uniform angles in 2-D, the Fibonacci sphere in 3-D
    (biapy/data/pre_processing.py:2058-2097 without jitter)."""
    if ndim == 2:
        a = np.linspace(0, 2 * np.pi, n_rays, endpoint=False, dtype=np.float32)
        return np.stack([np.cos(a), np.sin(a)], axis=1).astype(np.float32)
    k = np.arange(n_rays, dtype=np.float32)
    z = 1 - 2 * (k + 0.5) / n_rays
    rad = np.sqrt(np.maximum(0.0, 1 - z * z))
    theta = 2 * np.pi * k / ((1 + np.sqrt(5.0)) / 2.0)
    d = np.stack([rad * np.cos(theta), rad * np.sin(theta), z], axis=1).astype(np.float32)
    return d / (np.linalg.norm(d, axis=1, keepdims=True) + 1e-12)


def build_tta_spec(channel_names: Sequence[str], ndim: int, channel_extra_opts: Optional[Dict] = None, anisotropy: Optional[Sequence[float]] = None) -> "TTASpec":
    """The spec of ``build_tta_spec`` (tta.py:700-866) from the physical channel names: flows / HoVer maps -> signed vectors, ``E_offset_i`` /
    ``E_sigma_i`` (i in Cartesian x, y, z order) -> signed / unsigned vectors with the voxel spacing as their axis scale, ``R_k`` -> rays,
    ``A{z,y,x}_d`` -> affinities, everything else scalar.  Same group order and group names as the reference."""
    import re

    letters = _AXIS_LETTERS[ndim]
    names = list(channel_names)
    groups: List[object] = []
    taken = set()

    def axis_of(letter):
        return letters.index(letter) if letter in letters else None

    for family in ("flow", "hover"):
        comp = {nm: names.index(nm) for nm, (fam, _) in _COMPONENTS.items() if fam == family and nm in names}
        if not comp:
            continue
        per_axis: List[Optional[int]] = [None] * ndim
        for nm, ch in comp.items():
            ax = axis_of(_COMPONENTS[nm][1])
            taken.add(ch)
            if ax is None:                                     # a z component declared on 2-D data carries no in-plane direction
                groups.append(ScalarChannels(channels=(ch,)))
            else:
                per_axis[ax] = ch
        if any(c is not None for c in per_axis):
            groups.append(VectorChannels(axis_channels=tuple(per_axis), signed=True, name=family))
    scale = tuple(float(v) for v in anisotropy) if anisotropy is not None and len(anisotropy) == ndim else None
    for fam in ("E_offset", "E_sigma"):
        comps = {int(m.group(1)): i for i, nm in enumerate(names) if (m := re.match(r"^%s_(\d+)$" % fam, nm))}
        if not comps:
            continue
        per_axis = [None] * ndim
        for cart, ch in comps.items():
            ax = axis_of({0: "x", 1: "y", 2: "z"}.get(cart, "?"))
            taken.add(ch)
            if ax is None:
                groups.append(ScalarChannels(channels=(ch,)))
            else:
                per_axis[ax] = ch
        if any(c is not None for c in per_axis):
            groups.append(VectorChannels(axis_channels=tuple(per_axis), signed=(fam == "E_offset"), axis_scale=scale, name=fam))
    rays = sorted((int(m.group(1)), i) for i, nm in enumerate(names) if (m := re.match(r"^R_(\d+)$", nm)))
    if rays:
        pos = [i for _, i in rays]
        if pos != list(range(pos[0], pos[0] + len(pos))):
            raise ValueError("StarDist ray channels must be contiguous; got positions {}".format(pos))
        want = int((channel_extra_opts or {}).get("R", {}).get("nrays", len(pos)))
        if want != len(pos):
            raise ValueError("'R' declares nrays={} but {} ray output channels were found".format(want, len(pos)))
        d = np.asarray(_generate_rays(len(pos), ndim), dtype=np.float64)[:, ::-1].copy()        # Cartesian -> spatial-axis order
        d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-12
        groups.append(RayChannels(start=pos[0], dirs=d))
        taken.update(pos)
    layout: Dict[Tuple[int, int], int] = {}
    for i, nm in enumerate(names):
        m = re.match(r"^A([zyx])_(-?\d+)$", nm)
        if m:
            ax = axis_of(m.group(1))
            taken.add(i)
            if ax is None:
                groups.append(ScalarChannels(channels=(i,)))
            else:
                layout[(ax, int(m.group(2)))] = i
    if layout:
        groups.append(AffinityChannels(layout=layout))
    rest = tuple(i for i in range(len(names)) if i not in taken)
    if rest:
        groups.append(ScalarChannels(channels=rest))
    covered = sorted(c for g in groups for c in _group_channels(g))
    if covered != list(range(len(names))):
        raise ValueError("TTA spec does not cover every output channel exactly once (covered {} of {}); channel names were {}".format(len(covered), len(names), names))
    return TTASpec(ndim=ndim, n_channels=len(names), groups=groups)


def _group_channels(g) -> Tuple[int, ...]:
    k = _kind(g)
    if k == "vector":
        return tuple(c for c in g.axis_channels if c is not None)
    if k == "rays":
        return tuple(range(g.start, g.start + len(g.dirs)))
    if k == "affinities":
        return tuple(sorted(g.layout.values()))
    return tuple(g.channels)


def _inverse(perm, sign):
    """tta.py:125-137."""
    n = len(perm)
    pinv = [0] * n
    for a, p in enumerate(perm):
        pinv[p] = a
    return tuple(pinv), tuple(sign[pinv[b]] for b in range(n))


def _kind(g) -> str:
    """By the data a group carries, not by its ``name`` (the reference's build_tta_spec names vector groups after their family - "flow", "hover",
    "E_offset", "E_sigma" - tta.py:759, :786-793)."""
    if hasattr(g, "axis_channels"):
        return "vector"
    if hasattr(g, "dirs") and hasattr(g, "start"):
        return "rays"
    if hasattr(g, "layout"):
        return "affinities"
    if hasattr(g, "channels") and type(g).__name__ in ("ScalarChannels",):
        return "scalar"
    raise NotImplementedError(f"TTA channel group {type(g).__name__!r} is not known to biapy_amd.tta")


def _ray_permutation(dirs, perm, sign):
    """dest[j] = k with dirs[k] == t.inverse(dirs[j]) or None (tta.py:440-463)."""
    dirs = np.asarray(dirs)
    if len(dirs) == 0:
        return None
    pinv, sinv = _inverse(perm, sign)
    target = np.empty_like(dirs)
    for a in range(len(perm)):
        target[..., a] = sinv[a] * dirs[..., pinv[a]]
    dots = target @ dirs.T
    dest = np.argmax(dots, axis=1)
    if np.allclose(dots[np.arange(len(dest)), dest], 1.0, atol=1e-4) and len(np.unique(dest)) == len(dest):
        return dest.astype(np.int64)
    return None


def _supports(g, perm, sign) -> bool:
    """ChannelGroup.supports (tta.py:378-391, :465-471, :512-519)."""
    k = _kind(g)
    if k == "vector":
        pinv, _ = _inverse(perm, sign)
        ac, sc = g.axis_channels, g.axis_scale
        for a in range(len(perm)):
            src = pinv[a]
            if (ac[a] is None) != (ac[src] is None):
                return False
            if sc is not None and src != a and not np.isclose(sc[a], sc[src]):
                return False
        return True
    if k == "rays":
        return len(g.dirs) == 0 or _ray_permutation(g.dirs, perm, sign) is not None
    if k == "affinities":
        return all((perm[axis], off) in g.layout for (axis, off) in g.layout)
    return True


def filter_orientations(spec, orientations: Sequence[Orientation]) -> List[Orientation]:
    """TTASpec.filter_orientations (tta.py:589-621): the orientations every group can represent exactly; at least the identity."""
    kept = [(p, s) for p, s in orientations if all(_supports(g, p, s) for g in spec.groups)]
    n = len(orientations[0][0])
    return kept or [(tuple(range(n)), (1,) * n)]


def _mode_reducible(spec) -> List[int]:
    """TTASpec.mode_reducible_channels (tta.py:580-587): everything but signed vector components."""
    out: List[int] = []
    for g in spec.groups:
        k = _kind(g)
        if k == "scalar":
            out += list(g.channels)
        elif k == "vector":
            out += [] if g.signed else [c for c in g.axis_channels if c is not None]
        elif k == "rays":
            out += list(range(g.start, g.start + len(g.dirs)))
        else:
            out += sorted(g.layout.values())
    return sorted(out)


def _remap_channels(spec, pred: torch.Tensor, perm, sign) -> None:
    """TTASpec.remap_channels on a spatially restored device prediction (spatial..., C), in place (tta.py:393-402, :473-481, :521-544)."""
    n = len(perm)
    if tuple(perm) == tuple(range(n)) and all(v == 1 for v in sign):
        return
    if pred.shape[-1] != spec.n_channels:
        raise ValueError("TTA spec describes {} output channels but the model returned {}".format(spec.n_channels, pred.shape[-1]))
    pinv, sinv = _inverse(perm, sign)
    for g in spec.groups:
        k = _kind(g)
        if k == "vector":
            ac = g.axis_channels
            src = [pred[..., c].clone() if c is not None else None for c in ac]
            for a, dst in enumerate(ac):
                if dst is None:
                    continue
                comp = src[pinv[a]]
                pred[..., dst] = comp if (sinv[a] > 0 or not g.signed) else -comp
        elif k == "rays":
            dest = _ray_permutation(g.dirs, perm, sign)
            if dest is None:
                raise RuntimeError("remap called with an unsupported orientation")
            s0, nr = g.start, len(g.dirs)
            block = pred[..., s0:s0 + nr].clone()
            pred[..., (s0 + torch.from_numpy(dest)).to(pred.device)] = block
        elif k == "affinities":
            src = {key: pred[..., ch].clone() for key, ch in g.layout.items()}
            for (axis, off), block in src.items():
                dst_axis = perm[axis]
                dst = g.layout[(dst_axis, off)]
                if sign[axis] > 0:
                    pred[..., dst] = block
                else:
                    # a reversed axis turns offset +d into -d, and aff_{b,-d}(p) = aff_{b,+d}(p + d e_b): the map moves d voxels up the axis; the d
                    # leading slices that have no source repeat the first one (what seg2aff_pni's padding of the starting border amounts to)
                    n_b = block.shape[dst_axis]
                    if 0 < off < n_b:
                        head = block.narrow(dst_axis, 0, 1).expand(*[off if a == dst_axis else block.shape[a] for a in range(block.dim())])
                        pred[..., dst] = torch.cat([head, block.narrow(dst_axis, 0, n_b - off)], dim=dst_axis)
                    else:                                       # offsets outside the volume wrap around, as np.roll does in the reference
                        pred[..., dst] = torch.roll(block, shifts=off, dims=dst_axis)


def _is_scalar_only(spec) -> bool:
    return all(_kind(g) == "scalar" for g in spec.groups)


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
    spec = None if (tta_spec is None or _is_scalar_only(tta_spec)) else tta_spec      # an all-scalar spec IS the classic ensemble (tta.py:575-578)
    if not vol.is_cuda:
        raise RuntimeError("biapy_amd.tta runs on the MI355X only (volume is on %s); there is no CPU path" % vol.device)
    if vol.dim() != ndim + 1:
        raise ValueError("Expected a {}D input (spatial..., channels); got shape {}".format(ndim, tuple(vol.shape)))
    orientations = build_axis_transform_group(ndim, level=("full" if group == "auto" else group))
    if spec is not None:
        if spec.ndim != ndim:
            raise ValueError("TTA spec is {}D, the data {}D".format(spec.ndim, ndim))
        orientations = filter_orientations(spec, orientations)
    img = vol.to(torch.float32)
    # square off the axes that get swapped (post_processing.py:1285-1339): front padding, reflect (edge if too short)
    moved = sorted({a for p, _ in orientations for a in range(ndim) if p[a] != a} | {p[a] for p, _ in orientations for a in range(ndim) if p[a] != a})
    pad_before = None
    if moved:
        target = max(img.shape[a] for a in moved)
        if not all(img.shape[a] == target for a in moved):
            pad_before = [target - img.shape[a] if a in moved else 0 for a in range(ndim)]
            # reflections mirror the cells on the border and corrupt their flows / offsets / rays: zeros for a non-scalar spec (post_processing.py:1488-1491)
            pmode = "constant" if spec is not None else ("replicate" if any(pad_before[a] >= img.shape[a] for a in moved) else "reflect")
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
            if spec is not None:
                tmp = torch.empty_like(acc)
                acc_m = torch.empty_like(acc) if mode != "mean" else None
        for q, (perm, sign) in enumerate(chunk):
            k = b0 + q
            cp, cs, _ = _c3(perm, sign)
            if spec is not None:
                # generic path: un-orient into a scratch volume, remap the channels there, then reduce in orientation order (the sequential float32 sum
                # np.mean forms; min / max kept beside it for the mode-reducible channels)
                L.check(lib.bpx_tta_accumulate(pred[q].data_ptr(), Z, Y, X, Cout, cp, cs, 0, 1, 0, tmp.data_ptr(), st))
                _remap_channels(spec, tmp, perm, sign)
                if k == 0:
                    acc.copy_(tmp)
                    if acc_m is not None:
                        acc_m.copy_(tmp)
                else:
                    acc.add_(tmp)
                    if acc_m is not None:
                        (torch.minimum if mode == "min" else torch.maximum)(acc_m, tmp, out=acc_m)
                continue
            L.check(lib.bpx_tta_accumulate(pred[q].data_ptr(), Z, Y, X, Cout, cp, cs, {"mean": 0, "min": 1, "max": 2}[mode], 1 if k == 0 else 0,
                                           n_or if (mode == "mean" and k == n_or - 1) else 0, acc.data_ptr(), st))
    if spec is not None:
        acc = acc / float(n_or)                                       # np.mean: the float32 sum divided by n
        if mode != "mean":
            idx = _mode_reducible(spec)
            if idx:
                ii = torch.tensor(idx, dtype=torch.long, device=acc.device)
                acc[..., ii] = acc_m[..., ii]
    if pad_before is not None:
        acc = acc[tuple(slice(p, None) for p in pad_before) + (slice(None),)].contiguous()
    return acc
