"""Sliding-window inference on the device: crop -> forward -> blend, optionally sharded over the GPUs of a node.

Host-side counterpart of ``Base_Workflow.process_test_sample`` (per-patch branch,
biapy/engine/base_workflow.py:1944-1997) and ``predict_batches_in_test`` (:1696-1728):

    crop_3D_data_with_overlap(X, PATCH_SIZE, overlap, padding)      -> bpx_crop3d_gather, batch by batch
    model_call_func + apply_model_activations (sigmoid)              -> ResUNet engine, head activation fused
    merge_3D_data_with_overlap(pred, shape, padding, overlap)        -> bpx_merge3d_blend (deterministic gather)

Differences by design (MI355X-first, same results):
  * nothing returns to the host between the three stages: patches are gathered from the HBM-resident volume
    per batch (no 34 GB patch array), predictions stay in HBM (34 GB at cfg 3 - sized for 288 GB), and the
    blend reads them once;
  * multi-GPU: the reference runs this path on rank 0 only (base_workflow.py:1552-1559).  Here the patch grid
    is cut along Z into contiguous row slabs, one per rank; forwards are independent; the blend of the slices
    two slabs share is made bit-identical to the single-device result by handing the un-normalised partial sums
    (numerator, weight sum) of the boundary slices to the next rank, which SEEDS its accumulation with them
    (fp32 addition is not associative; the reference order is z-major, so lower ranks' terms come first).
    One neighbour send/recv per boundary + one all-gather (or gather to rank 0) of the disjoint output slabs,
    over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import tiling


def split_rows(n_rows: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [lo,hi) ranges of patch z-rows; ranks beyond n_rows get empty ranges."""
    base, rem = divmod(n_rows, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


@dataclass
class SlabPlan:
    """Which output slices a rank finalises / hands over, derived only from the integer grid."""
    rows: Tuple[int, int]          # patch z-rows [lo,hi) owned by the rank
    own: Tuple[int, int]           # output slices [z0,z1) the rank normalises and owns
    recv: Optional[Tuple[int, int]]  # slices whose partial sums arrive from the previous non-empty rank
    send: Optional[Tuple[int, int]]  # slices whose partial sums go to the next non-empty rank
    prev: Optional[int]
    next: Optional[int]


def plan_slabs(row_starts: Sequence[int], core_patch_z: int, Z: int, world: int) -> List[SlabPlan]:
    """Z-slab partition of the patch rows.  Needs non-decreasing row starts: the reference's placement rule shifts EVERY patch that
    would cross the volume end back by ``last`` (data_3D_manipulation.py:596-598), so with overlaps above 50 % on a volume of only
    a few patch lengths the starts can run backwards (e.g. 0,2,4,6,4,6,8) and a slice is no longer owned by consecutive ranks -
    such a volume is blended on one rank (it is tiny by construction)."""
    n_rows = len(row_starts)
    if world > 1 and any(row_starts[i] > row_starts[i + 1] for i in range(n_rows - 1)):
        raise ValueError(f"patch rows start at {list(row_starts)}: not monotonic, this geometry cannot be sharded by Z-slab (use world = 1)")
    ranges = split_rows(n_rows, world)
    active = [r for r in range(world) if ranges[r][1] > ranges[r][0]]
    plans: List[SlabPlan] = []
    for r in range(world):
        lo, hi = ranges[r]
        if hi <= lo:
            plans.append(SlabPlan((lo, hi), (0, 0), None, None, None, None))
            continue
        k = active.index(r)
        prev = active[k - 1] if k > 0 else None
        nxt = active[k + 1] if k + 1 < len(active) else None
        zs = 0 if prev is None else row_starts[lo]
        ze = row_starts[hi - 1] + core_patch_z                     # one past the last slice this rank's patches touch
        own_hi = Z if nxt is None else row_starts[ranges[nxt][0]]
        recv = None
        if prev is not None:
            prev_ze = row_starts[ranges[prev][1] - 1] + core_patch_z
            if prev_ze > zs:
                recv = (zs, min(prev_ze, Z))
        send = None
        if nxt is not None and ze > own_hi:
            send = (own_hi, min(ze, Z))
        plans.append(SlabPlan((lo, hi), (zs, own_hi), recv, send, prev, nxt))
    return plans


class DeviceBlend:
    """Blend backend on the MI355X kernels."""

    def __init__(self, plan: tiling.MergePlan):
        self.plan = plan

    def blend(self, patches, z_lo, z_hi, rows, acc=None, wacc=None, seed=False, write_partial=False, out_dtype=None, out=None):
        return tiling.merge_device(patches, self.plan, out_dtype=out_dtype, z_lo=z_lo, z_hi=z_hi, zrow_lo=rows[0], zrow_hi=rows[1],
                                   acc=acc, wacc=wacc, seed=seed, write_partial=write_partial, out=out)


def _peer(group, r: int) -> int:
    return dist.get_global_rank(group, r) if group is not None else r


def _exchange(ops):
    """One grouped launch of the given point-to-point operations (RCCL runs the sends and receives of a group concurrently, so
    a rank's send never queues behind its own receive); returns the requests."""
    return dist.batch_isend_irecv(ops) if ops else []


def _partial(n_slices: int, Y: int, X: int, C: int, dev, zero: bool):
    """Numerator and weight-sum partials of ``n_slices`` output slices in ONE buffer (one message per boundary)."""
    nvox = n_slices * Y * X
    buf = (torch.zeros if zero else torch.empty)(nvox * (C + 1), dtype=torch.float32, device=dev)
    return buf, buf[: nvox * C].view(n_slices, Y, X, C), buf[nvox * C:].view(n_slices, Y, X, 1)


def sharded_blend(backend, patches: torch.Tensor, plans: List[SlabPlan], rank: int, world: int, Y: int, X: int, C: int,
                  gather: str = "all", group=None) -> Optional[torch.Tensor]:
    """Blend this rank's predictions into its output slab, exchanging boundary partial sums with the neighbours.

    ``patches`` holds the predictions of the rank's patch rows only.  Returns the full volume (Z,Y,X,C) on every
    rank (gather="all"), on rank 0 only ("rank0"), or just the rank's own slab ("none").

    Order of work on a rank (the first three steps do not wait for anybody):
      1. the partial sums of the slices handed to the next rank - at cfg 3 they never depend on what the previous rank
         sends (68 received slices end before the 68 sent ones begin), so all boundaries of the node travel at once instead of
         as a chain through the ranks;
      2. ONE grouped send + receive (numerator and weight sums share a buffer: one message per boundary);
      3. the part of the own slab no neighbour contributes to, blended while the exchange is in flight;
      4. the seeded part, after the receive.
    The own slab is blended straight into its place in the result volume where the rank holds one (no staging copy); the
    slabs are disjoint Z-ranges of a (Z,Y,X,C) array, i.e. contiguous, so the final gather receives / broadcasts directly into
    views of that volume; gather="all" is ONE in-place all-gather of the slabs' common part plus a broadcast per longer slab (`gather_layout`).
    """
    me = plans[rank]
    dev = patches.device
    Z = max(p.own[1] for p in plans)
    active = me.rows[1] > me.rows[0]
    z0, z1 = me.own
    holds_full = world > 1 and (gather == "all" or (gather == "rank0" and rank == 0))
    full = torch.empty((Z, Y, X, C), dtype=torch.float32, device=dev) if holds_full else None
    slab = None
    if active:
        slab = full[z0:z1] if holds_full else torch.empty((z1 - z0, Y, X, C), dtype=torch.float32, device=dev)
        buf_in = acc = wacc = buf_out = sacc = swacc = None
        seed_hi = z0
        if me.recv is not None:
            buf_in, acc, wacc = _partial(me.recv[1] - me.recv[0], Y, X, C, dev, zero=False)
            seed_hi = me.recv[1]
        if me.send is not None:
            buf_out, sacc, swacc = _partial(me.send[1] - me.send[0], Y, X, C, dev, zero=True)
        dependent = me.recv is not None and me.send is not None and seed_hi > me.send[0]   # > 50 % overlap with one row per rank
        reqs = []
        if dependent:
            for r in _exchange([dist.P2POp(dist.irecv, buf_in, _peer(group, me.prev), group)] if world > 1 else []):
                r.wait()
            s0, s1 = me.send
            k = seed_hi - s0                                  # a previous rank's terms reach into what we hand over
            sacc[:k] = acc[s0 - me.recv[0]:seed_hi - me.recv[0]]
            swacc[:k] = wacc[s0 - me.recv[0]:seed_hi - me.recv[0]]
            backend.blend(patches, s0, seed_hi, me.rows, acc=sacc[:k], wacc=swacc[:k], seed=True, write_partial=True)
            if s1 > seed_hi:
                backend.blend(patches, seed_hi, s1, me.rows, acc=sacc[k:], wacc=swacc[k:], write_partial=True)
            if world > 1:
                reqs = _exchange([dist.P2POp(dist.isend, buf_out, _peer(group, me.next), group)])
        else:
            ops = []
            if me.send is not None:
                backend.blend(patches, me.send[0], me.send[1], me.rows, acc=sacc, wacc=swacc, write_partial=True)
                ops.append(dist.P2POp(dist.isend, buf_out, _peer(group, me.next), group))
            if me.recv is not None:
                ops.append(dist.P2POp(dist.irecv, buf_in, _peer(group, me.prev), group))
            if world > 1:
                reqs = _exchange(ops)
        e = min(max(seed_hi, z0), z1)                         # [z0, e) is seeded by the previous rank, [e, z1) is ours alone
        if z1 > e:
            backend.blend(patches, e, z1, me.rows, out=slab[e - z0:])
        for r in reqs:
            r.wait()
        if e > z0:
            backend.blend(patches, z0, e, me.rows, acc=acc[: e - z0], wacc=wacc[: e - z0], seed=True, out=slab[: e - z0])
    if gather == "none" or world == 1:
        return slab
    owners = [(r, p.own) for r, p in enumerate(plans) if p.rows[1] > p.rows[0] and p.own[1] > p.own[0]]
    if gather == "all":
        _all_gather_slabs(full, plans, rank, world, group)
        return full
    if rank == 0:
        ops = [dist.P2POp(dist.irecv, full[o[0]:o[1]], _peer(group, r), group) for r, o in owners if r != 0]
    else:
        ops = [dist.P2POp(dist.isend, slab, _peer(group, 0), group)] if slab is not None and slab.numel() else []
    for r in _exchange(ops):
        r.wait()
    return full


def gather_layout(plans: List[SlabPlan], world: int):
    """How the ranks' disjoint output slabs travel in ONE all-gather (north_star: "RCCL all-gather over xGMI for the stitched output").

    An all-gather moves equal pieces.  ``plan_slabs`` gives every rank but the last the same slab (cfg 3 on 8 ranks: 7 x 120 slices and one
    of 184 - the last rank also owns what its patches reach beyond the grid step), so the common part travels IN PLACE: when rank r's slab
    starts at r * piece, the slabs' first ``piece`` slices are consecutive pieces of the result volume and the collective's output is the
    volume itself (no staging, no copies); what is left of longer slabs ("tails": one 64-slice block at cfg 3) follows as broadcasts.
    Returns ("inplace", piece, tails) or, for layouts that do not line up (ranks without rows, unequal steps), ("padded", longest, None):
    every slab copied into a piece of the longest slab's size, gathered, and copied out."""
    own = [p.own if p.rows[1] > p.rows[0] else (0, 0) for p in plans]
    sizes = [o[1] - o[0] for o in own]
    piece = min(sizes)
    if piece > 0 and all(own[r][0] == r * piece for r in range(world)):
        tails = [(r, (own[r][0] + piece, own[r][1])) for r in range(world) if sizes[r] > piece]
        return "inplace", piece, tails
    return "padded", max(sizes), None


def _all_gather_slabs(full: torch.Tensor, plans: List[SlabPlan], rank: int, world: int, group=None) -> None:
    """Every rank's blended slab (already in its place in ``full`` on its owner) into ``full`` on every rank: one ``all_gather_into_tensor``
    (+ broadcasts of the tails, see ``gather_layout``).  Pure data movement: the gathered volume has the owners' bits."""
    kind, piece, tails = gather_layout(plans, world)
    Z = full.shape[0]
    row = full[0].numel()
    flat = full.view(-1)
    if kind == "inplace":
        z0 = plans[rank].own[0]
        mine = flat[z0 * row:(z0 + piece) * row]
        if dist.get_backend(group) != "nccl":
            mine = mine.clone()                    # RCCL takes the in-place form (input = its own piece of the output); other backends get a copy
        dist.all_gather_into_tensor(flat[: world * piece * row], mine, group=group)
        reqs = [dist.broadcast(flat[a * row:b * row], src=_peer(group, r), group=group, async_op=True) for r, (a, b) in tails]
        for q in reqs:
            q.wait()
        return
    me = plans[rank]
    active = me.rows[1] > me.rows[0]
    send = torch.zeros(piece * row, dtype=full.dtype, device=full.device)
    if active and me.own[1] > me.own[0]:
        send[: (me.own[1] - me.own[0]) * row] = flat[me.own[0] * row:me.own[1] * row]
    got = torch.empty(world * piece * row, dtype=full.dtype, device=full.device)
    dist.all_gather_into_tensor(got, send, group=group)
    for r, p in enumerate(plans):
        if r != rank and p.rows[1] > p.rows[0] and p.own[1] > p.own[0]:
            flat[p.own[0] * row:p.own[1] * row] = got[r * piece * row:r * piece * row + (p.own[1] - p.own[0]) * row]
    assert Z * row == flat.numel()


class SlidingWindowPredictor:
    """crop -> forward -> merge of one volume, entirely on the device(s)."""

    def __init__(self, model, patch_zyx: Sequence[int], overlap=(0.5, 0.5, 0.5), padding=(0, 0, 0), batch_size: int = 4,
                 pad_type: str = "reflect", forward: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 tta: Optional[str] = None, tta_mode: str = "mean", compute_dtype: Optional[torch.dtype] = None):
        """compute_dtype: storage type of the forward during ``predict`` (the model's own ``compute_dtype`` is restored afterwards);
        ``torch.float16`` is the inference mode whose Dice agrees with the fp32 reference to < 1e-4 at the speed of bf16 - the natural
        choice after bf16 training.  None = leave the model as it is.
        tta: None, or the orientation group of TEST.AUGMENTATION ("full" = 16 orientations in 3D, "flips" = 8): every patch
        is then predicted through ``biapy_amd.tta.ensemble_predictions`` (predict_batches_in_test, base_workflow.py:1659-1673)."""
        self.model = model
        self.compute_dtype = compute_dtype
        self.tta, self.tta_mode = tta, tta_mode
        self.patch = tuple(int(p) for p in patch_zyx)
        self.overlap, self.padding, self.batch, self.pad_type = tuple(overlap), tuple(padding), int(batch_size), pad_type
        # forward: (B,C,Z,Y,X) fp32 -> (B,Cout,Z,Y,X) fp32 probabilities (ce_sigmoid head)
        self.forward = forward or (lambda x: model.predict_proba(x))

    # ---- the steps of process_test_sample around the blended prediction (base_workflow.py:2089-2141; the padding itself is
    #      pad_to_shape, data_manipulation.py:3218-3300, called by the test generators with DATA.REFLECT_TO_COMPLETE_SHAPE) ----------------
    @staticmethod
    def _gather_axes(vol: torch.Tensor, idx_zyx) -> torch.Tensor:
        """out[z, y, x, :] = vol[iz[z], iy[y], ix[x], :] by the table-gather kernel (no PyTorch indexing on the volume)."""
        import numpy as np

        from . import _lib as L

        Z, Y, X, C = vol.shape
        tabs = torch.from_numpy(np.concatenate([np.asarray(t, dtype=np.int32) for t in idx_zyx])).to(vol.device)
        Pz, Py, Px = (len(t) for t in idx_zyx)
        out = torch.empty((Pz, Py, Px, C), dtype=vol.dtype, device=vol.device)
        L.check(L.lib.bpx_gather3d_tables(vol.data_ptr(), vol.element_size(), Z, Y, X, C, tabs.data_ptr(), 1, Pz, Py, Px, out.data_ptr(), L.stream_ptr()))
        return out

    def pad_to_shape(self, vol: torch.Tensor, mode: str = "reflect") -> torch.Tensor:
        """``pad_to_shape(img, crop_shape)`` of the reference: every spatial axis shorter than the patch is extended IN FRONT (the image stays in
        the bottom-right corner) with ``np.pad(..., mode)``; the source indices come from NumPy itself (np.pad of an index ramp), so repeated
        reflections of very short axes are the reference's own."""
        import numpy as np

        if mode != "reflect":
            raise NotImplementedError("pad_to_shape: only the reference's default mode 'reflect' is implemented on the device")
        idx = []
        for n, p in zip(vol.shape[:3], self.patch):
            ramp = np.arange(int(n))
            idx.append(np.pad(ramp, (p - int(n), 0), "reflect") if n < p else ramp)
        if all(len(t) == n for t, n in zip(idx, vol.shape[:3])):
            return vol
        return self._gather_axes(vol.contiguous(), idx)

    @torch.no_grad()
    def process_test_sample(self, vol: torch.Tensor, reflect_to_complete_shape: bool = True, class_channels: int = 0, **predict_kw) -> Optional[torch.Tensor]:
        """The per-patch branch of ``Base_Workflow.process_test_sample`` around ``predict`` (base_workflow.py:1874-2141) for a whole volume
        on this rank: DATA.REFLECT_TO_COMPLETE_SHAPE padding of axes shorter than the patch (what the test generator does before the
        workflow sees the sample), crop -> forward -> blend, the crop back to ``reflected_orig_shape`` (:2089-2131: the LAST n voxels of every
        axis), and the class head (:2135-2141, ``class_channels`` = the width of the separated class block: the trailing channels
        become one arg-max channel).  Returns (Z, Y, X, C') float32, or None on ranks that do not hold the gathered result."""
        from . import _lib as L

        assert vol.is_cuda and vol.dim() == 4 and vol.dtype == torch.float32
        orig = tuple(int(v) for v in vol.shape[:3])
        if predict_kw.get("full_z") is not None or predict_kw.get("z_offset", 0):
            raise ValueError("process_test_sample takes the WHOLE volume (its reflect completion and the crop back work on the full extent); "
                             "slab inputs (z_offset / full_z) go to predict() directly")
        work = self.pad_to_shape(vol) if reflect_to_complete_shape else vol
        pred = self.predict(work, **predict_kw)
        if pred is None:
            return None
        if tuple(pred.shape[:3]) != orig:
            import numpy as np

            pred = self._gather_axes(pred.contiguous(), [np.arange(p - n, p) for p, n in zip(pred.shape[:3], orig)])
        if class_channels:
            C = pred.shape[-1]
            out = torch.empty(pred.shape[:3] + (C - class_channels + 1,), dtype=torch.float32, device=pred.device)
            pred = pred.contiguous()
            L.check(L.lib.bpx_class_argmax(pred.data_ptr(), out.numel() // out.shape[-1], C, int(class_channels), out.data_ptr(), L.stream_ptr()))
            pred = out
        return pred

    def input_slab(self, vol_zyx: Sequence[int], rank: int, world: int):
        """Input slices [z_lo, z_hi) of a (Z,Y,X) volume that ``rank`` of ``world`` reads: the extent of its patch rows plus the
        reflect-padding sources at the volume ends ("each GPU reads only its input slab (+halo)").  Ranks without
        patch rows get (0, 0)."""
        Z, Y, X = (int(v) for v in vol_zyx)
        g = tiling.merge_grid((Z, Y, X), self.patch, self.overlap, self.padding)
        lo, hi = split_rows(g[0].n, world)[rank]
        if hi <= lo:
            return 0, 0
        return tiling.crop_rows_needed((Z, Y, X), self.patch, self.overlap, self.padding, lo, hi, reflect=self.pad_type != "zeros")

    @torch.no_grad()
    def predict(self, vol: torch.Tensor, rank: int = 0, world: int = 1, gather: str = "all", group=None, z_offset: int = 0,
                full_z: Optional[int] = None) -> Optional[torch.Tensor]:
        """vol: (Z,Y,X,C) float32 on this rank's device - the whole volume, or (``full_z`` given) only the slices
        [z_offset, z_offset + vol.shape[0]) of a volume of ``full_z`` slices, which must cover ``input_slab(...)`` of this rank.
        Returns the blended probability volume (Z,Y,X,Cout)."""
        assert vol.is_cuda and vol.dim() == 4 and vol.dtype == torch.float32
        if self.compute_dtype is not None and getattr(self.model, "compute_dtype", self.compute_dtype) != self.compute_dtype:
            from .engine import set_compute_dtype
            keep = set_compute_dtype(self.model, self.compute_dtype)
            try:
                return self.predict(vol, rank, world, gather, group, z_offset, full_z)
            finally:
                self.model.compute_dtype = keep
        Zs, Y, X, Cin = vol.shape
        Z = Zs if full_z is None else int(full_z)
        plan = tiling.MergePlan((Z, Y, X), self.patch, self.overlap, self.padding, vol.device)
        nz, ny, nx = plan.grid[0].n, plan.grid[1].n, plan.grid[2].n
        row_starts = [plan.row_start(i) for i in range(nz)]
        core_z = self.patch[0] - 2 * self.padding[0]
        plans = plan_slabs(row_starts, core_z, Z, world)
        lo, hi = plans[rank].rows
        n_mine = (hi - lo) * ny * nx
        pred = None
        for b0 in range(0, n_mine, self.batch):
            nb = min(self.batch, n_mine - b0)
            xb = tiling.crop_device(vol, self.patch, self.overlap, self.padding, self.pad_type, c_begin=lo * ny * nx + b0, c_count=nb,
                                    z_offset=z_offset, full_z=Z)
            if self.tta:
                from . import tta as _tta

                outs = [_tta.ensemble_predictions(xb[q], lambda b: self.forward(b.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1), 3,
                                                  batch_size_value=self.batch, mode=self.tta_mode, group=self.tta) for q in range(nb)]
                out5 = torch.stack(outs, 0)                                       # (nb, Z, Y, X, Cout)
            else:
                out5 = self.forward(xb.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1)   # to_pytorch_format / to_numpy_format (misc.py:689-733)
            if pred is None:
                pred = torch.empty((n_mine,) + self.patch + (out5.shape[-1],), dtype=torch.float32, device=vol.device)
            pred[b0:b0 + nb] = out5
        if pred is None:
            pred = torch.empty((0,) + self.patch + (1,), dtype=torch.float32, device=vol.device)
        cout = pred.shape[-1]
        if world > 1:  # every rank must agree on the channel count even when it owns no rows
            t = torch.tensor([cout], device=vol.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            cout = int(t.item())
        return sharded_blend(DeviceBlend(plan), pred, plans, rank, world, Y, X, cout, gather=gather, group=group)
