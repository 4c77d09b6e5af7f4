"""Dilated 3x3x3 convolutions on the ordinary conv kernels: space-to-batch / batch-to-space (ASPP, biapy/models/heads.py:77-104).

A convolution with dilation d couples only voxels whose coordinates agree modulo d, so it is d^3 independent ordinary
("same", zero-padded) convolutions on the sub-lattices x = r (mod d).  ``lattice_tables`` lists, for every residue r, the source
index of every sub-lattice voxel along z, y, x (-1 past the end of the volume: the zero padding that makes all sub-lattices the
same size); ``space_to_batch`` gathers them into the batch dimension with ``bpx_gather3d_tables``, ``batch_to_space`` scatters a
result back with ``bpx_scatter3d_tables``.  The same pair transports gradients (the transform is a permutation plus zero padding,
so its adjoint is the inverse), which is how dgrad / wgrad of a dilated convolution run on the existing kernels too.

``space_to_packed`` / ``packed_to_space`` are the form the ResUNet++ engine uses: the d^3 sub-lattices of a sample are laid out side by
side in ONE volume, separated by one plane of zeros along every axis (``packed_tables``).  A 3x3x3 kernel reaches one voxel, so the
zero planes are the "same" padding of every sub-lattice and ONE ordinary convolution of the packed volume is the d^3 independent
ones; its results on the separator planes are never read back.  Small sub-lattices (5^3 at rate 18 on an 80^3 volume) then fill the
convolution kernels' 4x8x16-voxel tiles instead of occupying a quarter of four tiles each: the packed volume is d*(n+1)-1 = 107 voxels
wide, 2.4x the voxels of the source against 8.2x for the tiles of 23,328 separate 5^3 volumes.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

lib = L.lib


def lattice_shape(dim_zyx: Sequence[int], d: int) -> Tuple[int, int, int]:
    return tuple(-(-int(n) // d) for n in dim_zyx)


def lattice_tables(dim_zyx: Sequence[int], d: int) -> np.ndarray:
    """int32 (d^3, nz+ny+nx): residue (rz, ry, rx) in z-major order; entry k of an axis = r + d*k, or -1 where that is outside."""
    n = lattice_shape(dim_zyx, d)
    rows = []
    for rz in range(d):
        for ry in range(d):
            for rx in range(d):
                t = []
                for r, nn, lim in zip((rz, ry, rx), n, dim_zyx):
                    idx = r + d * np.arange(nn, dtype=np.int64)
                    t.append(np.where(idx < lim, idx, -1).astype(np.int32))
                rows.append(np.concatenate(t))
    return np.stack(rows)


def space_to_batch(x: torch.Tensor, d: int, tables: torch.Tensor = None) -> torch.Tensor:
    """x (N, Z, Y, X, C) -> (N*d^3, nz, ny, nx, C); sample n, residue q sits at batch index n*d^3 + q."""
    if not x.is_cuda:
        raise RuntimeError("biapy_amd.dilation runs on the MI355X only; there is no CPU path")
    x = x.contiguous()
    N, Z, Y, X, C = x.shape
    nz, ny, nx = lattice_shape((Z, Y, X), d)
    if tables is None:
        tables = torch.from_numpy(lattice_tables((Z, Y, X), d)).to(x.device)
    q = d ** 3
    out = torch.empty((N * q, nz, ny, nx, C), dtype=x.dtype, device=x.device)
    per = out[0].numel() * q
    for n in range(N):
        L.check(lib.bpx_gather3d_tables(x[n].data_ptr(), x.element_size(), Z, Y, X, C, tables.data_ptr(), q, nz, ny, nx,
                                        out.data_ptr() + n * per * out.element_size(), L.stream_ptr()))
    return out


def batch_to_space(y: torch.Tensor, d: int, dim_zyx: Sequence[int], tables: torch.Tensor = None) -> torch.Tensor:
    """(N*d^3, nz, ny, nx, C) -> (N, Z, Y, X, C); the inverse of ``space_to_batch`` on the voxels inside the volume."""
    if not y.is_cuda:
        raise RuntimeError("biapy_amd.dilation runs on the MI355X only; there is no CPU path")
    y = y.contiguous()
    q = d ** 3
    Z, Y, X = (int(v) for v in dim_zyx)
    nz, ny, nx = lattice_shape((Z, Y, X), d)
    assert y.shape[0] % q == 0 and tuple(y.shape[1:4]) == (nz, ny, nx)
    N, C = y.shape[0] // q, y.shape[-1]
    if tables is None:
        tables = torch.from_numpy(lattice_tables((Z, Y, X), d)).to(y.device)
    out = torch.empty((N, Z, Y, X, C), dtype=y.dtype, device=y.device)
    per = y[0].numel() * q
    for n in range(N):
        L.check(lib.bpx_scatter3d_tables(y.data_ptr() + n * per * y.element_size(), y.element_size(), tables.data_ptr(), q, nz, ny, nx,
                                         out[n].data_ptr(), Z, Y, X, C, L.stream_ptr()))
    return out


def packed_shape(dim_zyx: Sequence[int], d: int) -> Tuple[int, int, int]:
    """Extent of the packed volume: d sub-lattices of n voxels and d-1 separator planes per axis."""
    return tuple(d * (n + 1) - 1 for n in lattice_shape(dim_zyx, d))


def packed_tables(dim_zyx: Sequence[int], d: int) -> np.ndarray:
    """int32 (1, pz+py+px): packed coordinate p = r*(n+1) + k holds source index r + d*k; separators and indices past the end are -1."""
    n = lattice_shape(dim_zyx, d)
    t = []
    for nn, lim in zip(n, dim_zyx):
        p = np.arange(d * (nn + 1) - 1, dtype=np.int64)
        r, k = p // (nn + 1), p % (nn + 1)
        idx = r + d * k
        t.append(np.where((k < nn) & (idx < lim), idx, -1).astype(np.int32))
    return np.concatenate(t)[None]


def space_to_packed(x: torch.Tensor, d: int, tables: torch.Tensor = None) -> torch.Tensor:
    """x (N, Z, Y, X, C) -> (N, pz, py, px, C) with zeros on the separator planes and past the end of the volume."""
    if not x.is_cuda:
        raise RuntimeError("biapy_amd.dilation runs on the MI355X only; there is no CPU path")
    x = x.contiguous()
    N, Z, Y, X, C = x.shape
    pz, py, px = packed_shape((Z, Y, X), d)
    if tables is None:
        tables = torch.from_numpy(packed_tables((Z, Y, X), d)).to(x.device)
    out = torch.empty((N, pz, py, px, C), dtype=x.dtype, device=x.device)
    for n in range(N):
        L.check(lib.bpx_gather3d_tables(x[n].data_ptr(), x.element_size(), Z, Y, X, C, tables.data_ptr(), 1, pz, py, px, out[n].data_ptr(),
                                        L.stream_ptr()))
    return out


def packed_to_space(y: torch.Tensor, d: int, dim_zyx: Sequence[int], tables: torch.Tensor = None) -> torch.Tensor:
    """(N, pz, py, px, C) -> (N, Z, Y, X, C): the inverse of ``space_to_packed`` on the voxels of the volume."""
    if not y.is_cuda:
        raise RuntimeError("biapy_amd.dilation runs on the MI355X only; there is no CPU path")
    y = y.contiguous()
    Z, Y, X = (int(v) for v in dim_zyx)
    pz, py, px = packed_shape((Z, Y, X), d)
    assert tuple(y.shape[1:4]) == (pz, py, px)
    N, C = y.shape[0], y.shape[-1]
    if tables is None:
        tables = torch.from_numpy(packed_tables((Z, Y, X), d)).to(y.device)
    out = torch.empty((N, Z, Y, X, C), dtype=y.dtype, device=y.device)
    for n in range(N):
        L.check(lib.bpx_scatter3d_tables(y[n].data_ptr(), y.element_size(), tables.data_ptr(), 1, pz, py, px, out[n].data_ptr(), Z, Y, X, C,
                                         L.stream_ptr()))
    return out
