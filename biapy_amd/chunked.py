"""By-chunks inference: padded, non-blended tiles.

Host-side mirror of the grid arithmetic of ``biapy/data/generators/chunked_test_pair_data_generator.py`` (``__init__``
:244-289, ``_patch_coords`` :440-487, ``extract_and_prepare_sample`` :524-565, ``__iter__``'s ``DistributedSampler`` :603-612)
and of the write-back of ``Base_Workflow.process_test_sample_by_chunks`` (``base_workflow.py:2573-2610``): the volume is cut
into chunks of ``PATCH - 2*PADDING`` voxels; every chunk is read with its padding, clipped to the volume and completed by
``np.pad(..., "reflect")`` to the patch size; its prediction is stored without the padding.  Chunks never overlap, so unlike
the spline-blended merge there is no arithmetic on the way back and the route is the reference's multi-GPU one: chunks are
dealt to the ranks in ``DistributedSampler`` order and each rank writes its own chunks.

Two predictors:

* ``ChunkedPredictor`` - the volume and the result live in HBM (288 GB hold a 4096^3 uint8 volume or a 2048^3 float32 one next to
  its prediction), the gather and the write-back are two HIP kernels (``bpx_gather3d_tables`` / ``bpx_scatter3d_regions``) and the ranks'
  disjoint results are combined by one RCCL reduction;
* ``StreamedChunkedPredictor`` (round 3) - OUT OF CORE, what by-chunks inference is for: the volume stays on the HOST side (any array-like
  with NumPy slicing: ``np.memmap`` of a raw / ``.npy`` file on disk, an in-memory array; ``zarr`` / ``h5py`` datasets have the same slicing
  interface but are not installed in this image), the prediction is written to a host-side array-like in CHUNK-ALIGNED regions (the
  reference's Zarr output is chunked by the write tile for the same reason: chunked_test_pair_data_generator.py:714-760), and the device
  only ever holds one work tile of ``patches_per_tile`` patches (``TEST.BY_CHUNKS.WORKFLOW_PROCESS`` tiles, :331-357) twice:
  pinned-memory staging, H2D of tile i + 1 and D2H of tile i - 1 on a copy stream while tile i computes.  The HBM footprint is
  bounded by the tile size, not by the volume.
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
        # overwrites identical data in the shared file - here a repeat (position >= total in the padded order) is left to the rank
        # that owns the chunk's first occurrence
        order = grid.rank_order(world, rank)
        mine = [v for k, v in enumerate(order) if rank + k * world < grid.total]
        # Several ranks, gather != "none": the ranks' results are disjoint sets of chunks, so they travel as ONE all-gather (round 5; an
        # all-reduce of whole volumes of zeros-elsewhere moved twice the bytes): every rank packs the cores (the patch without its padding, at
        # most `step` voxels per axis) of its chunks into slot k of a (slots, step...) buffer, the buffers are gathered, and one scatter places
        # every chunk of every rank - data movement only, the union bit for bit.
        packed = world > 1 and gather != "none"
        cz, cy, cx = grid.step
        slots = max(1, math.ceil(grid.total / world))
        cores = None
        seen = set()
        out = None
        slot_of = {}
        for k, v in enumerate(order):
            if rank + k * world < grid.total and v not in slot_of:
                slot_of[v] = k
        for b0 in range(0, len(mine), self.batch):
            ids = [v for v in mine[b0:b0 + self.batch] if v not in seen]   # chunks repeated to even out the ranks are predicted once
            seen.update(ids)
            if not ids:
                continue
            n = len(ids)
            tables = torch.from_numpy(np.stack([grid.index_tables(v) for v in ids])).to(vol.device, non_blocking=True)
            regs = np.stack([grid.region(v) for v in ids])
            if packed:   # destination: slot k of the core buffer, seen as a (slots * cz, cy, cx) volume
                regs = regs.copy()
                regs[:, 3] = [slot_of[v] * cz for v in ids]
                regs[:, 4:6] = 0
            regions = torch.from_numpy(regs).to(vol.device, non_blocking=True)
            patches = torch.empty((n, Pz, Py, Px, C), dtype=vol.dtype, device=vol.device)
            L.check(lib.bpx_gather3d_tables(vol.data_ptr(), vol.element_size(), Z, Y, X, C, tables.data_ptr(), n, Pz, Py, Px, patches.data_ptr(), st))
            pred = self.forward(patches.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1).contiguous().to(torch.float32)   # to_pytorch / to_numpy format
            Co = pred.shape[-1]
            if packed:
                if cores is None:
                    cores = torch.zeros((slots * cz, cy, cx, Co), dtype=torch.float32, device=vol.device)
                L.check(lib.bpx_scatter3d_regions(pred.data_ptr(), n, Pz, Py, Px, Co, regions.data_ptr(), cores.data_ptr(), slots * cz, cy, cx, st))
            else:
                if out is None:
                    out = torch.zeros((Z, Y, X, Co), dtype=torch.float32, device=vol.device)
                L.check(lib.bpx_scatter3d_regions(pred.data_ptr(), n, Pz, Py, Px, Co, regions.data_ptr(), out.data_ptr(), Z, Y, X, st))
        if packed:
            if cores is None:
                if self.out_channels is None:
                    raise RuntimeError("a rank without chunks cannot size the result: pass out_channels, or use world <= number of chunks")
                cores = torch.zeros((slots * cz, cy, cx, self.out_channels), dtype=torch.float32, device=vol.device)
            Co = cores.shape[-1]
            holds = gather == "all" or rank == 0
            got = torch.empty((world,) + tuple(cores.shape), dtype=torch.float32, device=vol.device) if holds else None
            if gather == "all":
                dist.all_gather_into_tensor(got.view(-1), cores.view(-1), group=group)
            else:   # reference semantics: rank 0 owns the result (base_workflow.py:1552-1559)
                dst = dist.get_global_rank(group, 0) if group is not None else 0
                dist.gather(cores, [got[r] for r in range(world)] if rank == 0 else None, dst=dst, group=group)
                if rank != 0:
                    return None
            # one scatter for the chunks of all ranks: "patch" (r, k) is slot k of rank r's buffer; empty slots have extent 0
            regs = np.zeros((world * slots, 9), dtype=np.int32)
            for r in range(world):
                done = set()
                for k, v in enumerate(grid.rank_order(world, r)):
                    if r + k * world < grid.total and v not in done:
                        done.add(v)
                        q = grid.region(v)
                        regs[r * slots + k, 3:] = q[3:]
            regions = torch.from_numpy(regs).to(vol.device, non_blocking=True)
            del cores   #  the rank's own cores are inside `got` now: the result volume takes their place instead of sitting beside them
            out = torch.zeros((Z, Y, X, Co), dtype=torch.float32, device=vol.device)
            # bpx_scatter3d_regions puts the patch index on grid.y (<= 65535): bounded groups of slots, in rank-major order like the one call was
            total, step_n = world * slots, 32768
            core_bytes = cz * cy * cx * Co * 4
            for s0 in range(0, total, step_n):
                n = min(step_n, total - s0)
                L.check(lib.bpx_scatter3d_regions(got.data_ptr() + s0 * core_bytes, n, cz, cy, cx, Co, regions.data_ptr() + s0 * 9 * 4, out.data_ptr(), Z, Y, X, st))
        return out


class TileGrid:
    """Work tiles of ``patches_per_tile`` consecutive chunks of the global grid (chunked_test_pair_data_generator.py:331-357): every chunk
    belongs to exactly one tile, tiles (not chunks) are dealt to the ranks (:608-624)."""

    def __init__(self, grid: ChunkGrid, patches_per_tile: Sequence[int] = (1, 1, 1)):
        self.grid = grid
        self.ppt = tuple(max(1, int(v)) for v in patches_per_tile)
        self.tiles = tuple(math.ceil(v / p) for v, p in zip(grid.vols, self.ppt))
        self.patches_of_tile = {}
        for vol_id in range(grid.total):
            z, y, x = (int(v) for v in np.unravel_index(vol_id, grid.vols))
            tid = int(np.ravel_multi_index((z // self.ppt[0], y // self.ppt[1], x // self.ppt[2]), self.tiles))
            self.patches_of_tile.setdefault(tid, []).append(vol_id)
        self.tile_ids = sorted(self.patches_of_tile)

    def rank_order(self, world: int, rank: int) -> List[int]:
        """Tiles of one rank in processing order: DistributedSampler(tile_ids, shuffle=False) without the repeats that even out the ranks
        (the reference predicts them twice and overwrites identical data)."""
        n = len(self.tile_ids)
        return [self.tile_ids[k] for k in range(rank, n, world)]

    def write_region(self, tile_id: int) -> PatchCoords:
        """The voxels the tile's chunks write (the union of their ``real_patch_in_data``): a box aligned to the chunk grid."""
        tz, ty, tx = (int(v) for v in np.unravel_index(tile_id, self.tiles))
        g = self.grid
        lo = [q * p * s for q, p, s in zip((tz, ty, tx), self.ppt, g.step)]
        hi = [min((q + 1) * p * s, d) for q, p, s, d in zip((tz, ty, tx), self.ppt, g.step, g.dim)]
        return PatchCoords(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])

    def read_region(self, tile_id: int) -> PatchCoords:
        """What the tile's chunks read: the write region grown by the padding, clipped to the volume."""
        w, g = self.write_region(tile_id), self.grid
        lo = [max(0, a - p) for a, p in zip(w[0::2], g.padding)]
        hi = [min(d, b + p) for b, p, d in zip(w[1::2], g.padding, g.dim)]
        return PatchCoords(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])


class StreamedChunkedPredictor:
    """Out-of-core by-chunks prediction: ``predict(vol_host, out_host)`` with both arrays on the HOST side (``np.memmap`` / ndarray / anything
    with NumPy basic slicing), any size; the device holds two work tiles.  Bit-identical to ``ChunkedPredictor`` on the same volume: a chunk's
    padded patch only draws on voxels of its own clipped read region, which lies inside its tile's read region, and the forward of a
    sample does not depend on what else is in the batch (InstanceNorm is per sample)."""

    def __init__(self, forward: Callable[[torch.Tensor], torch.Tensor], crop_zyx: Sequence[int], padding: Sequence[int], batch_size: int = 4,
                 patches_per_tile: Sequence[int] = (1, 1, 1), out_channels: int = 1, max_device_bytes: Optional[int] = None):
        self.forward, self.crop, self.padding, self.batch = forward, tuple(int(v) for v in crop_zyx[:3]), tuple(int(v) for v in padding), int(batch_size)
        self.ppt = tuple(int(v) for v in patches_per_tile)
        self.out_channels = int(out_channels)
        self.max_device_bytes = max_device_bytes
        self.device_bytes = 0          # bytes of the predictor's own device buffers (two tiles in, two tiles out, one batch of patches)

    def _tile_bytes(self, tg: TileGrid, C: int, esize: int):
        g = tg.grid
        rd = [min(d, p * s + 2 * pad) for d, p, s, pad in zip(g.dim, tg.ppt, g.step, g.padding)]
        wr = [min(d, p * s) for d, p, s in zip(g.dim, tg.ppt, g.step)]
        return rd[0] * rd[1] * rd[2] * C * esize, wr[0] * wr[1] * wr[2] * self.out_channels * 4

    @torch.no_grad()
    def predict(self, vol_host, out_host, device=None, rank: int = 0, world: int = 1):
        """vol_host: (Z, Y, X, C) array-like of uint8 / float16 / float32; out_host: writable (Z, Y, X, out_channels) float32 array-like.  Every rank
        writes the regions of its own tiles (chunk-aligned, disjoint between ranks: a shared file needs no further coordination); returns the
        number of tiles this rank wrote."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("StreamedChunkedPredictor computes on the MI355X only; there is no CPU path")
        Z, Y, X, C = (int(v) for v in vol_host.shape)
        if tuple(int(v) for v in out_host.shape) != (Z, Y, X, self.out_channels):
            raise ValueError(f"out_host must be {(Z, Y, X, self.out_channels)}, got {tuple(out_host.shape)}")
        grid = ChunkGrid((Z, Y, X), self.crop, self.padding)
        tg = TileGrid(grid, self.ppt)
        np_dtype = np.dtype(vol_host.dtype)
        tdtype = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float16): torch.float16, np.dtype(np.float32): torch.float32}.get(np_dtype)
        if tdtype is None:
            raise ValueError(f"volume dtype {np_dtype} is not supported (uint8, float16, float32)")
        in_bytes, out_bytes = self._tile_bytes(tg, C, np_dtype.itemsize)
        Pz, Py, Px = self.crop
        self.device_bytes = 2 * (in_bytes + out_bytes) + self.batch * Pz * Py * Px * (C * np_dtype.itemsize + self.out_channels * 4)
        if self.max_device_bytes is not None and self.device_bytes > self.max_device_bytes:
            raise MemoryError(f"work tiles need {self.device_bytes} bytes of HBM, the budget is {self.max_device_bytes}: reduce patches_per_tile / batch_size")
        # two slots: pinned staging + device buffers for the tile read region and the tile result
        slots = [dict(h_in=torch.empty(in_bytes, dtype=torch.uint8).pin_memory(), d_in=torch.empty(in_bytes, dtype=torch.uint8, device=device),
                      h_out=torch.empty(out_bytes, dtype=torch.uint8).pin_memory(), d_out=torch.empty(out_bytes, dtype=torch.uint8, device=device),
                      ready=torch.cuda.Event(), done=torch.cuda.Event(), out_ready=torch.cuda.Event(), pending=None) for _ in range(2)]
        h2d, d2h = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)      # separate queues: an upload never waits behind a download
        main = torch.cuda.current_stream(device)
        mine = tg.rank_order(world, rank)

        def stage_in(k):
            sl, rr = slots[k % 2], tg.read_region(mine[k])
            ext = (rr.z_end - rr.z_start, rr.y_end - rr.y_start, rr.x_end - rr.x_start)
            n = ext[0] * ext[1] * ext[2] * C
            if k >= 2:
                sl["ready"].synchronize()                                       # the upload that last read this pinned buffer has finished
            # the staging buffers were sized by _tile_bytes(): a read region beyond that bound must fail HERE, not as an opaque size error of a
            # truncated slice in the middle of a run
            assert n * np_dtype.itemsize <= in_bytes, f"read region {ext} of tile {mine[k]} exceeds the staging buffer ({n * np_dtype.itemsize} > {in_bytes} bytes)"
            # host read (disk for a memmap) straight into the pinned buffer through a NumPy view of it: no intermediate copy, and no
            # torch.from_numpy on a read-only memmap (a non-writable-array warning per tile)
            np.copyto(sl["h_in"][: n * np_dtype.itemsize].numpy().view(np_dtype).reshape(ext + (C,)),
                      vol_host[rr.z_start:rr.z_end, rr.y_start:rr.y_end, rr.x_start:rr.x_end].reshape(ext + (C,)))
            with torch.cuda.stream(h2d):
                h2d.wait_event(sl["done"])                                      # the tile that used this device buffer two tiles ago has been computed
                sl["d_in"][: n * np_dtype.itemsize].copy_(sl["h_in"][: n * np_dtype.itemsize], non_blocking=True)
                sl["ready"].record(h2d)
            sl["ext"], sl["rr"] = ext, rr

        def flush_out(sl):
            """The finished result of a slot: wait for its download, then write it to its chunk-aligned place in the host array."""
            if sl["pending"] is None:
                return
            wr, n = sl["pending"]
            sl["out_ready"].synchronize()
            ext = (wr.z_end - wr.z_start, wr.y_end - wr.y_start, wr.x_end - wr.x_start, self.out_channels)
            out_host[wr.z_start:wr.z_end, wr.y_start:wr.y_end, wr.x_start:wr.x_end] = sl["h_out"][: n * 4].view(torch.float32).reshape(ext).numpy()
            sl["pending"] = None

        for sl in slots:
            sl["done"].record(main)
            sl["out_ready"].record(main)
        if mine:
            stage_in(0)
        st = L.stream_ptr()
        for k, tid in enumerate(mine):
            sl = slots[k % 2]
            main.wait_event(sl["ready"])                                        # this tile's input is on the device
            main.wait_event(sl["out_ready"])                                    # the result that occupied this slot's device buffer has been downloaded
            rr, ext = sl["rr"], sl["ext"]
            wr = tg.write_region(tid)
            wext = (wr.z_end - wr.z_start, wr.y_end - wr.y_start, wr.x_end - wr.x_start)
            assert wext[0] * wext[1] * wext[2] * self.out_channels * 4 <= out_bytes, f"write region {wext} of tile {tid} exceeds the result buffer"
            tin = sl["d_in"][: ext[0] * ext[1] * ext[2] * C * np_dtype.itemsize].view(tdtype)
            tout = sl["d_out"][: wext[0] * wext[1] * wext[2] * self.out_channels * 4].view(torch.float32)
            ids = tg.patches_of_tile[tid]
            for b0 in range(0, len(ids), self.batch):
                sub = ids[b0:b0 + self.batch]
                n = len(sub)
                tabs = np.stack([grid.index_tables(v) for v in sub]).astype(np.int32)
                tabs[:, :Pz] -= rr.z_start
                tabs[:, Pz:Pz + Py] -= rr.y_start
                tabs[:, Pz + Py:] -= rr.x_start
                regs = np.stack([grid.region(v) for v in sub]).astype(np.int32)
                regs[:, 3] -= wr.z_start
                regs[:, 4] -= wr.y_start
                regs[:, 5] -= wr.x_start
                tables = torch.from_numpy(tabs).to(device, non_blocking=True)
                regions = torch.from_numpy(regs).to(device, non_blocking=True)
                patches = torch.empty((n, Pz, Py, Px, C), dtype=tdtype, device=device)
                L.check(lib.bpx_gather3d_tables(tin.data_ptr(), np_dtype.itemsize, ext[0], ext[1], ext[2], C, tables.data_ptr(), n, Pz, Py, Px, patches.data_ptr(), st))
                pred = self.forward(patches.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1).contiguous().to(torch.float32)
                if pred.shape[-1] != self.out_channels:
                    raise ValueError(f"the forward returned {pred.shape[-1]} channels, out_channels is {self.out_channels}")
                L.check(lib.bpx_scatter3d_regions(pred.data_ptr(), n, Pz, Py, Px, self.out_channels, regions.data_ptr(), tout.data_ptr(), wext[0], wext[1], wext[2], st))
            sl["done"].record(main)
            nout = wext[0] * wext[1] * wext[2] * self.out_channels
            with torch.cuda.stream(d2h):
                d2h.wait_event(sl["done"])
                sl["h_out"][: nout * 4].copy_(sl["d_out"][: nout * 4], non_blocking=True)          # the download overlaps the next tile's compute
                sl["out_ready"].record(d2h)
            sl["pending"] = (wr, nout)
            # while the device works on tile k: the previous tile's result goes to the host array, the next tile's input comes up
            other = slots[(k + 1) % 2]
            flush_out(other)
            if k + 1 < len(mine):
                stage_in(k + 1)
        for sl in slots:
            flush_out(sl)
        return len(mine)
