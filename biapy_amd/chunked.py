"""By-chunks inference: padded, non-blended tiles.

Host-side mirror of the grid arithmetic of ``biapy/data/generators/chunked_test_pair_data_generator.py`` (``__init__``
:244-289, ``_patch_coords`` :440-487, ``extract_and_prepare_sample`` :524-565, ``__iter__``'s ``DistributedSampler`` :603-612)
and of the write-back of ``Base_Workflow.process_test_sample_by_chunks`` (``base_workflow.py:2573-2610``): the volume is cut
into chunks of ``PATCH - 2*PADDING`` voxels; every chunk is read with its padding, clipped to the volume and completed by
``np.pad(..., "reflect")`` to the patch size; its prediction is stored without the padding.  Chunks never overlap, so unlike
the spline-blended merge there is no arithmetic on the way back and the route is the reference's multi-GPU one: chunks are
dealt to the ranks in ``DistributedSampler`` order and each rank writes its own chunks.

What is MI355X-specific: the volume and the result live in HBM (288 GB hold a 4096^3 uint8 volume or a 2048^3 float32
one next to its prediction), the gather and the write-back are two HIP kernels (``bpx_gather3d_tables`` /
``bpx_scatter3d_regions``) and the ranks' disjoint results are combined by one RCCL reduction instead of a Zarr file on disk.
The Zarr / HDF5 reading and writing of the reference (its on-disk format) and ``TEST.BY_CHUNKS.WORKFLOW_PROCESS`` tiles of
several patches are not part of this module.
"""
from __future__ import annotations

import math
from typing import Callable, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L

lib = L.lib


class PatchCoords(NamedTuple):
    z_start: int
    z_end: int
    y_start: int
    y_end: int
    x_start: int
    x_end: int


class ChunkGrid:
    """The chunk grid of one volume: ``vols_per_{z,y,x}`` chunks of ``step = crop - 2*padding`` voxels."""

    def __init__(self, vol_zyx: Sequence[int], crop_zyx: Sequence[int], padding: Sequence[int]):
        self.dim = tuple(int(v) for v in vol_zyx)
        self.crop = tuple(int(v) for v in crop_zyx[:3])
        self.padding = tuple(int(v) for v in padding)
        for ax, name in enumerate("ZYX"):
            if self.crop[ax] > self.dim[ax]:
                raise ValueError("{} Axis problem: {} greater than {} (you can reduce 'DATA.PATCH_SIZE' in that axis). Shape provided: {}".format(
                    name, self.crop[ax], self.dim[ax], self.dim))
        for i, p in enumerate(self.padding):
            if p >= self.crop[i] // 2:
                raise ValueError("'Padding' can not be greater than half of 'crop_shape'. Max value for the given input shape {} is {}".format(
                    self.crop, tuple(c // 2 - 1 for c in self.crop)))
        self.step = tuple(c - 2 * p for c, p in zip(self.crop, self.padding))
        self.vols = tuple(math.ceil(d / s) for d, s in zip(self.dim, self.step))
        self.total = self.vols[0] * self.vols[1] * self.vols[2]

    def patch_coords(self, vol_id: int) -> Tuple[int, int, int, PatchCoords, PatchCoords]:
        """(z, y, x) of the chunk, the region to read (padding included, clipped) and the region written back (:440-487)."""
        z, y, x = (int(v) for v in np.unravel_index(vol_id, self.vols))
        lo = [max(0, q * s - p) for q, s, p in zip((z, y, x), self.step, self.padding)]
        hi = [min((q + 1) * s + p, d) for q, s, p, d in zip((z, y, x), self.step, self.padding, self.dim)]
        rlo = [q * s for q, s in zip((z, y, x), self.step)]
        rhi = [min((q + 1) * s, d) for q, s, d in zip((z, y, x), self.step, self.dim)]
        return z, y, x, PatchCoords(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]), PatchCoords(rlo[0], rhi[0], rlo[1], rhi[1], rlo[2], rhi[2])

    def pad_to_add(self, vol_id: int) -> List[List[int]]:
        """Reflect padding that completes the clipped read region to the patch size, per axis [before, after] (:533-544)."""
        z, y, x, ext, _ = self.patch_coords(vol_id)
        out = []
        for q, s, p, c, a, b in zip((z, y, x), self.step, self.padding, self.crop, ext[0::2], ext[1::2]):
            left = abs(q * s - p) if q * s - p < 0 else 0
            out.append([left, c - (b - a) - left])
        return out

    def index_tables(self, vol_id: int) -> np.ndarray:
        """Source voxel index along z, y, x for every voxel of the patch, concatenated (length Pz+Py+Px, int32): the clipped read
        region followed by the same ``np.pad(..., "reflect")`` the reference applies to the data."""
        _, _, _, ext, _ = self.patch_coords(vol_id)
        pads = self.pad_to_add(vol_id)
        t = [np.pad(np.arange(a, b, dtype=np.int32), pw, "reflect") for a, b, pw in zip(ext[0::2], ext[1::2], pads)]
        return np.concatenate(t)

    def region(self, vol_id: int) -> np.ndarray:
        """{first kept voxel of the patch, destination, extent} (int32[9]): the prediction loses max(pad added, padding) voxels
        on every side (:557-562, base_workflow.py:2605-2609) and lands on the chunk's own region."""
        _, _, _, _, real = self.patch_coords(vol_id)
        pads = self.pad_to_add(vol_id)
        src = [max(pw[0], p) for pw, p in zip(pads, self.padding)]
        ext = [real.z_end - real.z_start, real.y_end - real.y_start, real.x_end - real.x_start]
        for ax in range(3):
            assert self.crop[ax] - src[ax] - max(pads[ax][1], self.padding[ax]) == ext[ax]
        return np.array(src + [real.z_start, real.y_start, real.x_start] + ext, dtype=np.int32)

    def rank_order(self, world: int, rank: int) -> List[int]:
        """Chunks of one rank, in processing order: ``DistributedSampler(tile_ids, num_replicas=world, rank=rank, shuffle=False)``
        (:603-612) - the id list is extended with its own head to a multiple of ``world`` and dealt round-robin."""
        ids = list(range(self.total))
        if not ids:
            return []
        total_size = math.ceil(len(ids) / world) * world
        pad = total_size - len(ids)
        ids += (ids * math.ceil(pad / len(ids)))[:pad]
        return ids[rank:total_size:world]


class ChunkedPredictor:
    """``predict(vol)``: by-chunks prediction of a device-resident ``(Z, Y, X, C)`` volume; returns ``(Z, Y, X, Cout)`` float32."""

    def __init__(self, forward: Callable[[torch.Tensor], torch.Tensor], crop_zyx: Sequence[int], padding: Sequence[int], batch_size: int = 4,
                 out_channels: Optional[int] = None):
        """out_channels: channels of the prediction; only needed when a rank can end up without chunks (world > chunks), so
        that it can still take part in the reduction with an all-zero partial result."""
        self.forward, self.crop, self.padding, self.batch = forward, tuple(int(v) for v in crop_zyx[:3]), tuple(int(v) for v in padding), int(batch_size)
        self.out_channels = out_channels

    @torch.no_grad()
    def predict(self, vol: torch.Tensor, rank: int = 0, world: int = 1, gather: str = "all", group=None) -> Optional[torch.Tensor]:
        if not vol.is_cuda:
            raise RuntimeError("ChunkedPredictor runs on the MI355X only (volume is on %s); there is no CPU path" % vol.device)
        if vol.dim() != 4:
            raise ValueError("volume must be (Z, Y, X, C)")
        vol = vol.contiguous()
        Z, Y, X, C = vol.shape
        grid = ChunkGrid((Z, Y, X), self.crop, self.padding)
        Pz, Py, Px = self.crop
        st = L.stream_ptr()
        # the sampler repeats head chunks so that every rank gets the same count; the reference predicts the repeats and
        # overwrites identical data in the shared file - here the ranks' results are SUMMED, so a repeat (position >= total in
        # the padded order) is left to the rank that owns the chunk's first occurrence
        mine = [v for k, v in enumerate(grid.rank_order(world, rank)) if rank + k * world < grid.total]
        seen = set()
        out = None
        for b0 in range(0, len(mine), self.batch):
            ids = [v for v in mine[b0:b0 + self.batch] if v not in seen]   # chunks repeated to even out the ranks are predicted once
            seen.update(ids)
            if not ids:
                continue
            n = len(ids)
            tables = torch.from_numpy(np.stack([grid.index_tables(v) for v in ids])).to(vol.device, non_blocking=True)
            regions = torch.from_numpy(np.stack([grid.region(v) for v in ids])).to(vol.device, non_blocking=True)
            patches = torch.empty((n, Pz, Py, Px, C), dtype=vol.dtype, device=vol.device)
            L.check(lib.bpx_gather3d_tables(vol.data_ptr(), vol.element_size(), Z, Y, X, C, tables.data_ptr(), n, Pz, Py, Px, patches.data_ptr(), st))
            pred = self.forward(patches.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1).contiguous().to(torch.float32)   # to_pytorch / to_numpy format
            if out is None:
                out = torch.zeros((Z, Y, X, pred.shape[-1]), dtype=torch.float32, device=vol.device)
            L.check(lib.bpx_scatter3d_regions(pred.data_ptr(), n, Pz, Py, Px, pred.shape[-1], regions.data_ptr(), out.data_ptr(), Z, Y, X, st))
        if world > 1 and gather != "none":   # "none": this rank's chunks only (zeros elsewhere), no collective
            # every chunk belongs to exactly one rank and the others hold zeros there: the sum is the union, bit for bit
            if out is None:
                if self.out_channels is None:
                    raise RuntimeError("a rank without chunks cannot size the result: pass out_channels, or use world <= number of chunks")
                out = torch.zeros((Z, Y, X, self.out_channels), dtype=torch.float32, device=vol.device)
            if gather == "all":
                dist.all_reduce(out, group=group)
            else:
                dist.reduce(out, dst=0, group=group)
                if rank != 0:
                    return None
        return out
