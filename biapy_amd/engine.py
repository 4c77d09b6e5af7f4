"""Executor of the 3D residual U-Net on the MI355X kernels (forward, and the hand-written backward).

This is the host side of biapy/models/resunet.py:352-446 (forward graph) and of the autograd graph
PyTorch would build for it (train_engine.py:173 ``loss.backward()``): it sequences the C-ABI kernels
of include/biapy_amd.h on the current HIP stream.  No arithmetic happens here and no PyTorch operator
touches an activation; PyTorch only owns the memory.

Data layout in HBM (all NDHWC, storage dtype T = bf16 or f32):
  * every raw (pre-normalisation) tensor is written once by its producer and read by its consumers;
    InstanceNorm+ELU never materialise - the producer emits per-(n,c) partial statistics, a tiny
    finalize kernel turns them into (mean, rstd, scale, shift) records and the consumer conv applies
    them while staging its input halo in LDS;
  * torch.cat([up, skip], 1) (blocks.py:1653) is a layout decision, not a kernel: each level owns one
    buffer [B,S,S,S,Cup+Cskip]; the transposed conv writes channels [0,Cup), the encoder block writes
    its output directly into [Cup, Cup+Cskip);
  * the residual add and the 1x1x1 shortcut conv (blocks.py:1372,1458) are extra K-steps of the
    block's second 3x3x3 conv kernel.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import ctypes as C

import os

import torch

from . import _lib as L

lib = L.lib
EPS = 1e-5

# Packed-operand caches are keyed by (storage address, tensor version) - but a HIP-graph replay of an optimizer step rewrites the
# parameters WITHOUT bumping ``_version``.  Every replaying object of this package (graphs.GraphedTrainStep /
# DataParallelTrainStep, train_engine._restore) therefore bumps this process-wide counter, which is part of every cache key.
_WEIGHTS_EPOCH = [0]


def bump_weights_epoch() -> None:
    _WEIGHTS_EPOCH[0] += 1


def weights_epoch() -> int:
    return _WEIGHTS_EPOCH[0]


@dataclass
class NetConfig:
    in_ch: int
    feature_maps: Sequence[int]
    out_channels: Sequence[int] = (1,)
    activation: str = "elu"
    normalization: str = "in"
    z_down: Optional[Sequence[int]] = None    # MODEL.Z_DOWN per level (1 or 2; None = 2 everywhere); YX_DOWN is always 2
    ndim: int = 3                             # 2: (B,C,Y,X) tensors, run as one-z-slice volumes (z_down is then 1 everywhere)
    post_up: int = 0                          # super-resolution "post" up-sampling ConvTranspose3d(fm0, fm0, k = s = (post_up, 2, 2)) in front
    #                                           of the heads (resunet.py:326-333, :399-400); 0 = none, else the z factor (1 or 2)
    dropout: Optional[Sequence[float]] = None # MODEL.DROPOUT_VALUES: one probability per level, the last one for the bottleneck (resunet.py:250, :270, :299); None / 0 = none

    def __post_init__(self):
        fm = list(self.feature_maps)
        zd = [2] * (len(fm) - 1) if self.z_down is None else [int(v) for v in list(self.z_down)[: len(fm) - 1]]
        if len(zd) != len(fm) - 1 or any(v not in (1, 2) for v in zd):
            raise NotImplementedError(f"z_down={self.z_down!r}: one value per level, each 1 or 2")
        self.z_down = tuple(zd)
        if self.normalization not in ("in", "gn"):
            raise NotImplementedError(f"normalization={self.normalization!r}: the MI355X engine implements 'in' (the reference default) and 'gn'")
        # 'gn': torch.nn.GroupNorm(8, C) for every norm layer - what blocks.py:2122-2125 means (the reference's own call,
        # nn.GroupNorm(out_channels, num_groups=8), raises a TypeError, so there is no reference output to pin this mode to)
        self.gn_groups = 8 if self.normalization == "gn" else 0
        if self.gn_groups:
            # checked here, not by a kernel in the middle of the first forward: bpx_norm_finalize / bpx_norm_bwd_finalize take
            # 1, 2, 4, ... 64 channels per group for the single-producer tensors; the concatenated decoder inputs (any channels per group) go
            # through the general group kernels, which only need whole groups
            ok_cpg = (1, 2, 4, 8, 16, 32, 64)
            for i, c in enumerate(fm):
                if c % self.gn_groups or (c // self.gn_groups) not in ok_cpg:
                    raise NotImplementedError(f"normalization='gn': feature_maps[{i}] = {c} gives {c / self.gn_groups:g} channels per group; the kernels "
                                              f"take {ok_cpg} (GroupNorm(8, C): C in 8 ... 512, a power of two times 8)")
            for i in range(len(fm) - 1):
                if (fm[i] + fm[i + 1]) % self.gn_groups:
                    raise NotImplementedError(f"normalization='gn': the concatenated decoder input of level {i} has {fm[i] + fm[i + 1]} channels, not a multiple of 8")
        if self.activation not in L.ACT:
            raise NotImplementedError(f"activation={self.activation!r} is not implemented on the MI355X engine")
        # Widths that are not multiples of 16 (e.g. MODEL.FEATURE_MAPS [52, 68, 84] of the reference's CartoCell template) run zero-padded to the next
        # multiple (the MFMA tile): `feature_maps` becomes what the kernels see, `true_feature_maps` what the parameters have.  Exact under InstanceNorm:
        # a padded channel is produced by zero weights and read through zero weights (channel_pad_plan / pad_channels below).
        self.true_feature_maps = None
        if any(c % 16 for c in fm):
            if self.gn_groups or self.post_up or self.in_ch != 1:
                raise NotImplementedError(f"feature_maps {fm} (not multiples of 16) run zero-padded: InstanceNorm, one input channel and no super-resolution stage only")
            self.true_feature_maps = tuple(fm)
            fm = [(c + 15) // 16 * 16 for c in fm]
            self.feature_maps = fm
        if self.in_ch != 1 and self.in_ch % 16:
            raise NotImplementedError("input channels must be 1 or a multiple of 16")
        if self.post_up not in (0, 1, 2):
            raise NotImplementedError("post up-sampling: z factor 1 or 2 (y / x factor 2)")
        if sum(self.out_channels) > 4:
            raise NotImplementedError("output head supports <= 4 channels")
        self.depth = len(fm) - 1
        dv = [0.0] * len(fm) if self.dropout is None else [float(v) for v in list(self.dropout)]
        if len(dv) < len(fm) or any(not (0.0 <= v < 1.0) for v in dv):
            raise ValueError(f"dropout={self.dropout!r}: one probability in [0, 1) per level and one for the bottleneck")
        self.dropout = tuple(dv[: len(fm) - 1]) + (dv[-1],)      # levels 0 .. depth-1, bottleneck (the reference indexes drop_values[i] and [-1])


def set_compute_dtype(model, dtype: torch.dtype) -> torch.dtype:
    """Switch the storage type of a drop-in model (workflow.SlidingWindowPredictor, train_engine.evaluate); returns the previous one.
    Refused up front - not by a kernel in the middle of a predict call - when the model's engine has no kernels for it."""
    ok = getattr(model, "supported_compute_dtypes", (torch.float32, torch.bfloat16))
    if dtype not in ok:
        raise NotImplementedError(f"{type(model).__name__}: compute_dtype={dtype} is not implemented for this model (supported: {', '.join(str(d) for d in ok)})")
    keep = model.compute_dtype
    model.compute_dtype = dtype
    return keep


def needs_lift(w: torch.Tensor) -> bool:
    return w.dim() == 4 or (w.dim() == 5 and w.shape[2] == 1 and w.shape[-1] == 3)


def lift_params(P: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Parameters in the shapes the 3-D kernels take.  Conv2d (Cout,Cin,3,3) and anisotropic Conv3d (Cout,Cin,1,3,3) weights
    become (Cout,Cin,3,3,3) with only the centre z-tap non-zero; other 4-D weights (1x1 convs, ConvTranspose2d (Cin,Cout,2,2))
    get a unit z extent (a view).  Everything else passes through."""
    Q = {}
    for k, w in P.items():
        if w.dim() == 4 and w.shape[-1] == 3:
            w5 = torch.zeros(w.shape[:2] + (3, 3, 3), dtype=torch.float32, device=w.device)
            w5[:, :, 1] = w
            Q[k] = w5
        elif w.dim() == 5 and w.shape[2] == 1 and w.shape[-1] == 3:
            w5 = torch.zeros(w.shape[:2] + (3, 3, 3), dtype=torch.float32, device=w.device)
            w5[:, :, 1] = w[:, :, 0]
            Q[k] = w5
        elif w.dim() == 4:
            Q[k] = w.reshape(w.shape[:2] + (1,) + w.shape[2:])
        else:
            Q[k] = w
    return Q


def unlift_grads(G: Dict[str, torch.Tensor], P: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Gradients of lifted parameters back in the shapes of the module's parameters (centre z-tap / dropped unit extent)."""
    out = {}
    for n, p in P.items():
        g = G[n]
        if g.shape != p.shape:
            g = g[:, :, 1].reshape(p.shape) if p.shape[-1] == 3 else g.reshape(p.shape)
        out[n] = g if g.is_contiguous() else g.contiguous()
    return out


def channel_pad_plan(cfg: "NetConfig") -> Optional[Dict[str, Tuple]]:
    """For a configuration whose widths were rounded up to multiples of 16: parameter name -> (segments of dim 0, segments of dim 1), a segment =
    (true channels, padded channels); dim 1 is None for vectors.  The decoder blocks' input is the concatenation [up-sampled | skip]: two segments."""
    if cfg.true_feature_maps is None:
        return None
    ft, fp, Lv = list(cfg.true_feature_maps), list(cfg.feature_maps), cfg.depth
    seg = lambda i: (ft[i], fp[i])                                         # noqa: E731
    plan: Dict[str, Tuple] = {}

    def block(prefix, first, cin, cout):
        k = block_keys(prefix, first)
        plan[k["w1"]], plan[k["wsc"]] = ([cout], cin), ([cout], cin)
        plan[k["w2"]] = ([cout], [cout])
        for n in ("b1", "g1", "be1", "b2", "bsc"):
            plan[k[n]] = ([cout], None)
        if not first:
            plan[k["g0"]], plan[k["be0"]] = (cin, None), (cin, None)

    for i in range(Lv):
        block(f"down_path.{i}", i == 0, [(cfg.in_ch, cfg.in_ch)] if i == 0 else [seg(i - 1)], seg(i))
    block("bottleneck", False, [seg(Lv - 1)], seg(Lv))
    for j, i in enumerate(range(Lv - 1, -1, -1)):
        plan[f"up_paths.0.{j}.up.weight"] = ([seg(i + 1)], [seg(i + 1)])   # ConvTranspose: (Cin, Cout, ...)
        plan[f"up_paths.0.{j}.up.bias"] = ([seg(i + 1)], None)
        block(f"up_paths.0.{j}.conv_block", False, [seg(i + 1), seg(i)], seg(i))
    for h in range(len(cfg.out_channels)):
        plan[f"heads.{h}.weight"] = (None, [seg(0)])
    return plan


def _pad_dim(t: torch.Tensor, dim: int, segs) -> torch.Tensor:
    if segs is None or all(a == b for a, b in segs):
        return t
    parts, o = [], 0
    for a, b in segs:
        parts.append(t.narrow(dim, o, a))
        o += a
        if b > a:
            shape = list(t.shape)
            shape[dim] = b - a
            parts.append(t.new_zeros(shape))
    return torch.cat(parts, dim)


def _unpad_dim(t: torch.Tensor, dim: int, segs) -> torch.Tensor:
    if segs is None or all(a == b for a, b in segs):
        return t
    parts, o = [], 0
    for a, b in segs:
        parts.append(t.narrow(dim, o, a))
        o += b
    return torch.cat(parts, dim) if len(parts) > 1 else parts[0]


def pad_channels(P: Dict[str, torch.Tensor], plan) -> Dict[str, torch.Tensor]:
    """Parameters with their channel dimensions zero-padded to the kernels' widths (channel_pad_plan)."""
    Q = {}
    for k, w in P.items():
        s0, s1 = plan.get(k, (None, None))
        Q[k] = _pad_dim(_pad_dim(w, 0, s0), 1, s1).contiguous()
    return Q


def unpad_channel_grads(G: Dict[str, torch.Tensor], plan) -> Dict[str, torch.Tensor]:
    """The parameters' own entries of the padded gradients (the padded rows / columns hold gradients of weights that do not exist)."""
    out = {}
    for k, g in G.items():
        s0, s1 = plan.get(k, (None, None))
        out[k] = _unpad_dim(_unpad_dim(g, 0, s0), 1, s1)
    return out


def block_keys(prefix: str, first: bool) -> Dict[str, str]:
    """Reference state_dict keys of one ResConvBlock."""
    i = 0 if first else 2
    k = {
        "w1": f"{prefix}.block.{i}.block.0.weight", "b1": f"{prefix}.block.{i}.block.0.bias",
        "g1": f"{prefix}.block.{i}.block.1.weight", "be1": f"{prefix}.block.{i}.block.1.bias",
        "w2": f"{prefix}.block.{i + 1}.block.0.weight", "b2": f"{prefix}.block.{i + 1}.block.0.bias",
        "wsc": f"{prefix}.shortcut.0.weight", "bsc": f"{prefix}.shortcut.0.bias",
    }
    if not first:
        k["g0"] = f"{prefix}.block.0.weight"
        k["be0"] = f"{prefix}.block.0.bias"
    return k


class _Stats:
    """Partial-statistics scratch + finalize into a norm-record array."""

    @staticmethod
    def alloc(B, tiles, C, dev):
        return torch.empty((B, tiles, 2, C), dtype=torch.float32, device=dev)

    @staticmethod
    def finalize(part, B, tiles, C, count, gamma, beta, rec, rec_ld, rec_off, st, groups=0):
        """InstanceNorm records (groups = 0 -> one group per channel) or GroupNorm(groups) records of ONE producer's tensor."""
        L.check(lib.bpx_norm_finalize(part.data_ptr(), B, tiles, C, count, gamma.data_ptr(), beta.data_ptr(), EPS, groups or C,
                                      rec.data_ptr(), rec_ld, rec_off, st))

    @staticmethod
    def finalize_cat(parts, B, count, gamma, beta, rec, groups, st):
        """GroupNorm(groups) records of torch.cat(producers, 1): parts = [(partials, tiles, C), ...] in channel order.  A group may straddle
        the boundary between two producers (8 groups over 48 channels = 6 per group), so the per-channel totals of all of them are
        gathered first (bpx_norm_channel_sums) and one more kernel forms the group statistics (bpx_groupnorm_finalize)."""
        Ct = sum(c for _, _, c in parts)
        sums = torch.empty((B, Ct, 2), dtype=torch.float64, device=rec.device)
        off = 0
        for part, tiles, c in parts:
            L.check(lib.bpx_norm_channel_sums(part.data_ptr(), B, tiles, c, sums.data_ptr(), Ct, off, st))
            off += c
        L.check(lib.bpx_groupnorm_finalize(sums.data_ptr(), B, Ct, count, gamma.data_ptr(), beta.data_ptr(), EPS, groups, rec.data_ptr(), st))


# InstanceNorm-backward finalize of the residual blocks: per-sample blocks + dgamma / dbeta with the deferred reductions (BPX_NBF_DEFER=0: the plain entry)
_NBF = lib.bpx_norm_bwd_finalize_deferred if os.environ.get("BPX_NBF_DEFER", "1") != "0" else lib.bpx_norm_bwd_finalize


def _recs(B, C, dev):
    return torch.empty((B, C, 4), dtype=torch.float32, device=dev)


@dataclass
class _Blk:
    """Everything the backward needs about one residual block."""
    keys: Dict[str, str]
    first: bool
    S: Tuple[int, int, int]
    cin: int
    cout: int
    x: Optional[torch.Tensor] = None      # raw block input buffer (NDHWC) - None for the first block (image)
    x_c0: int = 0                          # channel offset / count of the input view inside x
    rec_x: Optional[torch.Tensor] = None
    h: Optional[torch.Tensor] = None
    rec_h: Optional[torch.Tensor] = None
    out: Optional[torch.Tensor] = None     # buffer holding the block output
    out_c0: int = 0
    drop_p: float = 0.0                    # dropout of the block (blocks.py:163), active in training mode only
    site: int = 0                          # index of the block's dropout site (its own random stream)
    a: Optional[torch.Tensor] = None       # with dropout: the materialised act(norm(h)) * keep / (1 - p) the second convolution read
    drop_ctr: Optional[torch.Tensor] = None  # with dropout: the device step counter value THIS forward drew its masks with (its backward regenerates them from it)


class ResUNetEngine:
    def __init__(self, cfg: NetConfig, dtype: torch.dtype = torch.bfloat16):
        assert dtype in (torch.bfloat16, torch.float32, torch.float16)
        self.cfg = cfg
        self.dtype = dtype
        self._pad_plan = channel_pad_plan(cfg) if type(self) is ResUNetEngine else None      # zero-padded widths: this engine's own forward / backward only
        if cfg.true_feature_maps is not None and type(self) is not ResUNetEngine:
            raise NotImplementedError(f"feature_maps {list(cfg.true_feature_maps)} (not multiples of 16): only the ResUNet engine pads them")
        if type(self) is not ResUNetEngine and list(cfg.feature_maps)[0] not in (16, 32):
            raise NotImplementedError("output head supports 16 or 32 features (the GEMM-fed head for wider first levels is the ResUNet engine's)")
        # float16 = the same 16 bits per element with an 11-bit mantissa: the mode whose forward agrees with the fp32 reference to Dice
        # delta < 1e-4 at the speed of the bf16 mode.  Its TRAINING form is mixed (BPX_MIX16): the forward pass and every stored
        # activation are fp16, every gradient tensor and the backward MFMA operands are bf16 (fp32 exponent range: no loss scaling) -
        # the backward kernels read the fp16 activations and convert on the way to their bf16 operands.
        self.dt = {torch.bfloat16: L.BF16, torch.float32: L.F32, torch.float16: L.F16}[dtype]
        mixed = dtype == torch.float16
        self.gdtype = torch.bfloat16 if mixed else dtype         # storage type of gradient tensors
        self.gdt = L.BF16 if mixed else self.dt                  # dtype code of kernels that touch gradient tensors only
        self.bdt = L.MIX16 if mixed else self.dt                 # dtype code of backward kernels that also read a forward activation
        self.pdt = L.MIX16 if mixed else self.dt                 # weight packing: forward operators in the forward type, transposed ones in bf16
        self.act = L.ACT[cfg.activation]
        self._ws: Optional[torch.Tensor] = None
        self._side_stream = None
        self.planar_cat = os.environ.get("BPX_PLANAR_CAT", "1") != "0"
        self.use_side_stream = False  # ALL weight-gradient kernels on a second stream: measured on cfg 2 19.5 vs 18.1 ms/step (early kernels, eager);
        # 11.13 vs 11.18 with the final kernels under graph replay, 21 vs 15 ms eager - the big layers already launch one resident wave of workgroups.
        # Round 4: the same for the SMALL levels only (<= side_small_vps voxels per sample, e.g. 4096 = the 16^3 and 8^3 levels of cfg 2, whose kernels are
        # latency chains with one workgroup or less per CU) as a parallel branch of the captured graph: measured SLOWER too - same box, 30 graph-replayed
        # steps: off 9.74 ms, <= 16^3 9.84 ms, <= 32^3 9.91 ms - and left off (BPX_SIDE_VPS=<voxels> switches it on).
        self.side_small_vps = int(os.environ.get("BPX_SIDE_VPS", "0"))
        # dropout (p > 0 levels, training mode): see _drop_args
        self.drop_active = False               # set by the module before every forward (= module.training)
        self.drop_seed: Optional[int] = None
        self._drop_counter: Optional[torch.Tensor] = None
        self.drop_mask_io: Optional[Dict[int, torch.Tensor]] = None   # tests: {site: uint8 keep flags}; drop_mask_mode 1 = use them, 2 = record the drawn ones
        self.drop_mask_mode = 0
        self._side_used = False
        self._pack_cache: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self._pack_versions: Dict[Tuple[int, int, int], Tuple[int, int]] = {}

    def clear_caches(self) -> None:
        """Drop the packed / lifted weight copies kept for inference (ResUNet.train() calls this)."""
        self._pack_cache.clear()
        self._pack_versions.clear()
        self._lift_cache = None

    # ---- weight-gradient side stream ------------------------------------------------------------------
    # The wgrad kernels only produce parameter gradients; nothing on the dgrad chain waits for them.  They run on a second
    # HIP stream so that their workgroups co-reside with the dgrad kernels' (both are latency-bound at 2-3 waves/SIMD on
    # their own).  Ordering: side waits for an event recorded on the main stream when the kernel's inputs exist; the
    # main stream waits for the side stream once, at the end of backward.  Buffers read by side-stream kernels are kept
    # alive in ctx["keep"] until then (PyTorch's allocator is stream-ordered per stream, not across streams).
    def _side(self, dev, S=None):
        """The side stream for a launch at spatial size S (None: only in the all-launches mode), or None = the current stream."""
        small = S is not None and S[0] * S[1] * S[2] <= self.side_small_vps
        if not (self.use_side_stream or small):
            return None
        if self._side_stream is None or self._side_stream.device != dev:
            self._side_stream = torch.cuda.Stream(device=dev)
        return self._side_stream

    def _run_side(self, dev, fn, S=None):
        side = self._side(dev, S)
        if side is None:
            fn(L.stream_ptr())
            return
        self._side_used = True
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        side.wait_event(ev)
        with torch.cuda.stream(side):
            fn(side.cuda_stream)

    def _workspace(self, nbytes: int, dev) -> torch.Tensor:
        """Scratch for the wgrad partial sums.  Deferred reductions (backward): one slab per call, kept until the flush.
        Otherwise grow-only (launches are stream-ordered, so one slab is enough)."""
        if getattr(self, "_deferred", False):
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
            self._keep.append(ws)
            return ws
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            if self._ws is not None and hasattr(self, "_keep"):
                self._keep.append(self._ws)   # a side-stream kernel may still be using the old slab
            self._ws = torch.empty(max(nbytes, 32 << 20), dtype=torch.uint8, device=dev)
        return self._ws

    # ---- dropout --------------------------------------------------------------------------------------
    # nn.Dropout(p) of a block sits behind Conv -> Norm -> Act of its first ConvBlock (blocks.py:163), i.e. on the tensor the second convolution's
    # fused prologue would form on the fly.  For p > 0 in training mode that tensor is materialised with the mask applied
    # (bpx_norm_act_dropout_fwd), the second convolution runs without a prologue, and its backward is the plain dgrad followed by
    # bpx_norm_act_dropout_bwd (mask, activation derivative, IN-backward sums).  The mask is a counter-based function of (seed, step counter, site,
    # element): nothing is stored, a replayed graph draws a new mask at every step because the counter lives on the device.
    def _drop_state(self, dev):
        if self.drop_seed is None:
            self.drop_seed = int(torch.initial_seed()) & ((1 << 63) - 1)
        if self._drop_counter is None or self._drop_counter.device != dev:
            self._drop_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        return self._drop_counter

    def _drop_mask(self, blk: "_Blk", numel: int, dev):
        """(pointer, mode) of the test hook's keep-flag buffer for this site."""
        if not self.drop_mask_mode:
            return None, 0
        if self.drop_mask_io is None:
            self.drop_mask_io = {}
        m = self.drop_mask_io.get(blk.site)
        if m is None:
            assert self.drop_mask_mode == 2, f"dropout site {blk.site}: no mask given"
            m = self.drop_mask_io[blk.site] = torch.zeros(numel, dtype=torch.uint8, device=dev)
        assert m.numel() == numel and m.dtype == torch.uint8 and m.is_cuda
        return m.data_ptr(), self.drop_mask_mode

    def _wgrad(self, B, S, x: "L.Tensor", rec, act, dy: "L.Tensor", k, dw, db, st, dev, db2=None):
        """db2: a second bias gradient that takes the same sums (the shortcut bias of a residual block)."""
        D, H, W = S
        nb = lib.bpx_conv3d_wgrad_workspace(B, D, H, W, x.C, dy.C, k)
        ws = self._workspace(nb, dev)
        self._run_side(dev, lambda s_: L.check(lib.bpx_conv3d_wgrad_db2(self.bdt, B, D, H, W, x, L.ptr(rec), act, dy, k, dw.data_ptr(), L.ptr(db),
                                                                        L.ptr(db2), ws.data_ptr(), ws.numel(), s_)), S)

    def _bwd_fused_ok(self, B, S, Ct: int, Cdy: int) -> bool:
        """One-pass dgrad + wgrad (bpx_conv3d_bwd_fused) for this conv?  Not with the weight-gradient side stream (its point is one staging)."""
        if os.environ.get("BPX_FUSED_BITS") is not None and not getattr(self, "_fused_bits_set", False):      # A/B aid: bpx_debug_set_bwd_fused bits
            lib.bpx_debug_set_bwd_fused(int(os.environ["BPX_FUSED_BITS"]))
            self._fused_bits_set = True
        return (os.environ.get("BPX_BWD_FUSED", "1") != "0" and self.act <= 3 and not self.use_side_stream and self.cfg.gn_groups == 0 and self.dtype != torch.float32
                and bool(lib.bpx_conv3d_bwd_fused_supported(self.bdt, B, S[0], S[1], S[2], Ct, Cdy)))

    def _bwd_fused(self, B, S, dy: "L.Tensor", wt, t: "L.Tensor", rec, g: "L.Tensor", dw, db, db2, st, dev):
        D, H, W = S
        tiles = lib.bpx_conv3d_bwd_fused_stats_tiles(B, D, H, W, t.C, dy.C)
        red = torch.empty((B, tiles, 2, t.C), dtype=torch.float32, device=dev)
        ws = self._workspace(lib.bpx_conv3d_bwd_fused_workspace(B, D, H, W, t.C, dy.C), dev)
        L.check(lib.bpx_conv3d_bwd_fused(self.bdt, B, D, H, W, dy, wt.data_ptr(), t, rec.data_ptr(), self.act, g, red.data_ptr(),
                                         dw.data_ptr(), L.ptr(db), L.ptr(db2), ws.data_ptr(), ws.numel(), st))
        return tiles, red

    # ------------------------------------------------------------------------------------------
    def _pack_plan(self, train: bool):
        """(parameter name, pack mode, Cin, Cout) of every packed operand one step needs (forward; + backward if train)."""
        cfg = self.cfg
        fm, Lv = list(cfg.feature_maps), cfg.depth
        plan = []

        def block(prefix, first, cin, cout):
            k = block_keys(prefix, first)
            if not (first and cfg.in_ch == 1):
                plan.append((k["w1"], L.PK_K3, cin, cout))
                plan.append((k["wsc"], L.PK_K1, cin, cout))
            plan.append((k["w2"], L.PK_K3, cout, cout))
            if train:
                plan.append((k["w2"], L.PK_K3_T, cout, cout))
                if not first:
                    plan.append((k["w1"], L.PK_K3_T, cin, cout))
                    plan.append((k["wsc"], L.PK_DENSE_T, cin, cout))

        for i in range(Lv):
            block(f"down_path.{i}", i == 0, cfg.in_ch if i == 0 else fm[i - 1], fm[i])
        block("bottleneck", False, fm[Lv - 1], fm[Lv])
        for j, i in enumerate(range(Lv - 1, -1, -1)):
            cup = fm[i + 1]
            plan.append((f"up_paths.0.{j}.up.weight", L.PK_CT if cfg.z_down[i] == 2 else L.PK_CT4, cup, cup))
            if train:
                plan.append((f"up_paths.0.{j}.up.weight", L.PK_CT_T if cfg.z_down[i] == 2 else L.PK_CT4_T, cup, cup))
            block(f"up_paths.0.{j}.conv_block", False, cup + fm[i], fm[i])
        return plan

    def _prepack(self, P: Dict[str, torch.Tensor], train: bool, dev, plan=None) -> None:
        """Pack every MFMA weight operand of the step with ONE launch into one buffer (the weights change after every
        optimizer step, so training re-packs ~60 small tensors per step).  plan: the (name, mode, Cin, Cout) list to use instead of
        this engine's own (the ResUNet++ engine records its list on the first step)."""
        self._prepacked = {}
        if plan is None:
            plan = self._pack_plan(train)
        if any(P[name].dtype != torch.float32 or not P[name].is_contiguous() for name, _, _, _ in plan):
            return  # _pack() falls back to per-tensor packing (with the conversion copy)
        sizes = [int(lib.bpx_packed_weight_elems(mode, cin, cout, self.dt)) for _, mode, cin, cout in plan]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 127) // 128 * 128          # keep every operand 256-byte aligned
        buf = torch.empty(tot, dtype=self.dtype, device=dev)
        jobs = (L.PackJob * len(plan))()
        es = buf.element_size()
        for q, ((name, mode, cin, cout), off) in enumerate(zip(plan, offs)):
            w = P[name]
            jobs[q] = L.PackJob(w.data_ptr(), buf.data_ptr() + off * es, mode, cin, cout, 0)
            self._prepacked[(w.data_ptr(), mode)] = buf[off:off + sizes[q]]
        L.check(lib.bpx_pack_weights_batched(self.pdt, len(plan), C.cast(jobs, C.c_void_p), L.stream_ptr()))

    def _begin_recorded_packs(self, P: Dict[str, torch.Tensor], train: bool, dev, cache_weights: bool) -> None:
        """For the tape engines (ResUNet++, RCAN), whose list of packed operands is not written down: the first step of a kind
        (inference / training) packs its weights one by one and ``_pack`` records which PARAMETERS were packed how; every later step
        packs that list with one launch up front (78 launches -> 1 per ResUNet++ training step, ~820 -> 1 for the RCAN trunk)."""
        if not hasattr(self, "_pack_plans"):
            self._pack_plans = {}
        self._pack_names = {v.data_ptr(): k for k, v in P.items()}
        self._pack_seen = self._pack_plans.setdefault(bool(train), {})
        self._prepacked = {}
        if self._pack_seen and not cache_weights:
            self._prepack(P, train, dev, plan=list(self._pack_seen))

    def _pack(self, w: torch.Tensor, mode: int, cin: int, cout: int, cache: bool) -> torch.Tensor:
        pre = getattr(self, "_prepacked", None)
        if pre:
            hit = pre.get((w.data_ptr(), mode))
            if hit is not None:
                return hit
        names = getattr(self, "_pack_names", None)
        if names:
            name = names.get(w.data_ptr())
            if name is not None:
                self._pack_seen[(name, mode, cin, cout)] = True       # a parameter (not a per-step temporary): part of the next step's batch
        key = (w.data_ptr(), mode, self.dt)
        stamp = (w._version, _WEIGHTS_EPOCH[0])
        if cache and key in self._pack_cache and self._pack_versions.get(key) == stamp:
            return self._pack_cache[key]
        n = lib.bpx_packed_weight_elems(mode, cin, cout, self.dt)
        out = torch.empty(n, dtype=self.dtype, device=w.device)
        wc = w.detach()
        if wc.dtype != torch.float32 or not wc.is_contiguous():
            wc = wc.float().contiguous()
        L.check(lib.bpx_pack_weight(mode, wc.data_ptr(), cin, cout, self.pdt, out.data_ptr(), L.stream_ptr()))
        if cache:
            self._pack_cache[key] = out
            self._pack_versions[key] = stamp
        return out

    # ------------------------------------------------------------------------------------------
    def _res_block_fwd(self, P, blk: _Blk, B, img: Optional[torch.Tensor], st, cache: bool, want_out_stats: bool, pool=None):
        """Runs one residual block. Returns partial stats (part, tiles) of the block output if requested.
        pool = (z stride, pooled tensor, its partial-statistics tensor): fuse the MaxPool3d that follows an encoder block into
        the epilogue of the block's last convolution (bpx_conv3d_fwd_pool; the caller checked that it is supported)."""
        D, H, W = blk.S
        dev = blk.out.device
        k = blk.keys
        vox = D * H * W
        C1 = blk.cout
        # ---- conv1 -> h (+ stats) ----------------------------------------------------------------
        if blk.first and self.cfg.in_ch == 1:
            tiles = lib.bpx_conv3d_c1_stats_tiles(D, H, W)
            part = _Stats.alloc(B, tiles, C1, dev)
            L.check(lib.bpx_conv3d_c1_fwd(self.dt, B, D, H, W, img.data_ptr(), P[k["w1"]].data_ptr(), P[k["b1"]].data_ptr(),
                                          L.tview(blk.h), part.data_ptr(), st))
        else:
            tiles = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, C1)
            part = _Stats.alloc(B, tiles, C1, dev)
            wp = self._pack(P[k["w1"]], L.PK_K3, blk.cin, C1, cache)
            L.check(lib.bpx_conv3d_fwd(self.dt, B, D, H, W, L.tview(blk.x, blk.x_c0, blk.cin),
                                       L.ptr(blk.rec_x), self.act if blk.rec_x is not None else 0, wp.data_ptr(), P[k["b1"]].data_ptr(),
                                       L.NULL_T, None, None, L.tview(blk.h), part.data_ptr(), st))
        blk.rec_h = _recs(B, C1, dev)
        _Stats.finalize(part, B, tiles, C1, vox, P[k["g1"]], P[k["be1"]], blk.rec_h, C1, 0, st, self.cfg.gn_groups)
        # ---- dropout: conv2 reads the materialised, masked activation instead of forming it in its prologue -------
        x2, rec2, act2 = L.tview(blk.h), blk.rec_h.data_ptr(), self.act
        if blk.drop_p > 0.0 and self.drop_active:
            # THIS forward's counter value (a device copy taken in forward(): capture-safe), kept with the block so that the backward of this pass
            # regenerates this pass's masks even when another forward ran in between (two forwards before one backward, retained graphs)
            ctr = blk.drop_ctr = self._drop_ctr_cur
            blk.a = torch.empty_like(blk.h)
            mptr, mmode = self._drop_mask(blk, blk.h.numel(), dev)
            L.check(lib.bpx_norm_act_dropout_fwd(self.dt, B, vox, L.tview(blk.h), blk.rec_h.data_ptr(), self.act, blk.drop_p, self.drop_seed, ctr.data_ptr(),
                                                 blk.site, mptr, mmode, L.tview(blk.a), st))
            x2, rec2, act2 = L.tview(blk.a), None, 0
        # ---- conv2 (+ shortcut, + residual add) -> out (+ stats) ----------------------------------
        wp2 = self._pack(P[k["w2"]], L.PK_K3, C1, C1, cache)
        tiles2 = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, C1)
        part2 = _Stats.alloc(B, tiles2, C1, dev) if want_out_stats else None
        if blk.first and self.cfg.in_ch == 1:
            sc = L.Tensor(img.data_ptr(), 1, 1)
            wsc_ptr = P[k["wsc"]].data_ptr()  # (Cout,1,1,1,1) fp32 used as a vector
        else:
            sc = L.tview(blk.x, blk.x_c0, blk.cin)
            wsc_ptr = self._pack(P[k["wsc"]], L.PK_K1, blk.cin, C1, cache).data_ptr()
        if pool is not None:
            L.check(lib.bpx_conv3d_fwd_pool(self.dt, B, D, H, W, x2, rec2, act2, wp2.data_ptr(),
                                            P[k["b2"]].data_ptr(), sc, wsc_ptr, P[k["bsc"]].data_ptr(),
                                            L.tview(blk.out, blk.out_c0, C1), L.ptr(part2), pool[0], L.tview(pool[1]), pool[2].data_ptr(), st))
        else:
            L.check(lib.bpx_conv3d_fwd(self.dt, B, D, H, W, x2, rec2, act2, wp2.data_ptr(),
                                       P[k["b2"]].data_ptr(), sc, wsc_ptr, P[k["bsc"]].data_ptr(),
                                       L.tview(blk.out, blk.out_c0, C1), L.ptr(part2), st))
        return part2, tiles2

    # ------------------------------------------------------------------------------------------
    def forward(self, P: Dict[str, torch.Tensor], x: Optional[torch.Tensor], head_act: int = 0, save: bool = False, cache_weights: bool = False,
                x_ndhwc: Optional[torch.Tensor] = None, want_dx: bool = False):
        """x: (B,C,Z,Y,X) fp32 with channels_last_3d strides (or any layout for C == 1).  Returns logits
        (B,sum(out_ch),Z,Y,X) fp32 in channels-first planar layout, and the saved context (or None).
        ``x_ndhwc``: the input already as a dense (B,Z,Y,X,C) tensor of the storage dtype (``x`` is then ignored); ``want_dx``: the
        backward also returns the gradient of that tensor under the key "__dx__" (super-resolution pre-up-sampling, resunet_sr)."""
        cfg = self.cfg
        if x_ndhwc is not None:
            assert x_ndhwc.is_cuda and x_ndhwc.dtype == self.dtype and x_ndhwc.dim() == 5 and x_ndhwc.is_contiguous() and cfg.in_ch != 1
            x = x_ndhwc.permute(0, 4, 1, 2, 3)           # only its shape is used below
        else:
            assert x.is_cuda and x.dtype == torch.float32 and x.dim() == cfg.ndim + 2
        if cfg.ndim == 2:
            x = x.unsqueeze(2)
        if self.drop_active and any(v > 0 for v in cfg.dropout):
            # a new mask per forward pass (a device value: replayed graphs advance it too); the pass works from its own snapshot
            self._drop_ctr_cur = self._drop_state(x.device).add_(1).clone()
        P_orig = P
        if cache_weights and torch.cuda.is_current_stream_capturing():
            # a captured forward must contain its own pack kernels: operands cached during the warm-up would freeze the
            # weights of every later replay at their capture-time values (graphs.GraphedInference)
            cache_weights = False
        plan = self._pad_plan
        if plan is not None or any(needs_lift(w) for w in P.values()):
            # 2D / anisotropic levels: zero-padded 3x3x3 weights.  Inference keeps the lifted copies while the parameters
            # are unchanged, so that the packed-operand cache (keyed by storage) keeps hitting.
            vers = (_WEIGHTS_EPOCH[0],) + tuple((w.data_ptr(), w._version) for w in P.values())
            hit = getattr(self, "_lift_cache", None)
            if cache_weights and hit is not None and hit[0] == vers:
                P = hit[1]
            else:
                P = lift_params(P if plan is None else pad_channels(P, plan))
                self._lift_cache = (vers, P) if cache_weights else None
        B, Cin, D0, H0, W0 = x.shape
        assert Cin == cfg.in_ch, f"expected {cfg.in_ch} input channels, got {Cin}"
        Lv = cfg.depth
        div = 2 ** Lv
        zdiv = 1
        for v in cfg.z_down:
            zdiv *= v
        if D0 % zdiv or H0 % div or W0 % div:
            raise ValueError(f"patch {D0, H0, W0} must be divisible by {(zdiv, div, div)} (DATA.PATCH_SIZE rule, check_configuration.py:3156-3202)")
        dev = x.device
        st = L.stream_ptr()
        fm = list(cfg.feature_maps)
        T = self.dtype
        if not cache_weights:
            self._prepack(P, save, dev)       # ONE batched pack launch (training re-packs every step; so does a captured inference)
        else:
            self._prepacked = {}
        if Cin == 1:
            img = x.reshape(B, D0, H0, W0).contiguous()
            x_ndhwc = None
        elif x_ndhwc is not None:
            img = None
        else:
            img = None
            xin = x.permute(0, 2, 3, 4, 1).contiguous()
            x_ndhwc = torch.empty(xin.shape, dtype=T, device=dev)
            if T == torch.float32:
                x_ndhwc.copy_(xin)
            else:
                L.check(lib.bpx_cast(L.F32, xin.data_ptr(), self.dt, x_ndhwc.data_ptr(), xin.numel(), st))

        S = [(D0, H0, W0)]
        for i in range(Lv):
            S.append((S[i][0] // cfg.z_down[i], S[i][1] // 2, S[i][2] // 2))

        def buf(i, C):
            return torch.empty((B,) + S[i] + (C,), dtype=T, device=dev)

        # torch.cat([up, skip], 1) buffers, chunk-planar (L.Planar): the transposed conv and the encoder block write whole planes
        # instead of 64 / 32 of every 96 bytes, pooling reads a dense plane (A/B switch: self.planar_cat)
        cat = [L.Planar(B, S[i], fm[i + 1] + fm[i], T, dev) if self.planar_cat else buf(i, fm[i + 1] + fm[i]) for i in range(Lv)]
        blocks: List[_Blk] = []
        pools = []
        out_stats = []
        cur, cur_rec = x_ndhwc, None
        # ---------------- encoder ------------------------------------------------------------------
        for i in range(Lv):
            blk = _Blk(keys=block_keys(f"down_path.{i}", i == 0), first=(i == 0), S=S[i], cin=(cfg.in_ch if i == 0 else fm[i - 1]),
                       cout=fm[i], x=cur, x_c0=0, rec_x=cur_rec, h=buf(i, fm[i]), out=cat[i], out_c0=fm[i + 1], drop_p=cfg.dropout[i], site=i)
            # pool -> P_i (+ stats) and the pre-norm record of the next block; fused into the block's last conv where the
            # lean kernel runs (>= 64^3 levels, bf16): the output slice is then not read again
            pooled = buf(i + 1, fm[i])
            D, H, W = S[i]
            k_ = blk.keys
            fused = bool(lib.bpx_conv3d_fwd_pool_supported(self.dt, B, D, H, W, fm[i], cat[i].shape[-1], fm[i])) and all(
                P[k_[q]].data_ptr() % 16 == 0 for q in ("b2", "bsc", "wsc"))
            if fused:
                ptiles = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, fm[i])
                ppart = _Stats.alloc(B, ptiles, fm[i], dev)
                part, tiles = self._res_block_fwd(P, blk, B, img, st, cache_weights, want_out_stats=True, pool=(cfg.z_down[i], pooled, ppart))
            else:
                part, tiles = self._res_block_fwd(P, blk, B, img, st, cache_weights, want_out_stats=True)
                ptiles = lib.bpx_maxpool3d_stats_tiles(self.dt, D, H, W, cfg.z_down[i], fm[i])
                ppart = _Stats.alloc(B, ptiles, fm[i], dev)
                L.check(lib.bpx_maxpool3d_fwd(self.dt, B, D, H, W, cfg.z_down[i], L.tview(cat[i], fm[i + 1], fm[i]), L.tview(pooled), ppart.data_ptr(), st))
            out_stats.append((part, tiles))
            blocks.append(blk)
            nxt = "bottleneck" if i == Lv - 1 else f"down_path.{i + 1}"
            rec = _recs(B, fm[i], dev)
            _Stats.finalize(ppart, B, ptiles, fm[i], S[i + 1][0] * S[i + 1][1] * S[i + 1][2], P[f"{nxt}.block.0.weight"],
                            P[f"{nxt}.block.0.bias"], rec, fm[i], 0, st, cfg.gn_groups)
            pools.append(pooled)
            cur, cur_rec = pooled, rec
        # ---------------- bottleneck ----------------------------------------------------------------
        bot = _Blk(keys=block_keys("bottleneck", False), first=False, S=S[Lv], cin=fm[Lv - 1], cout=fm[Lv], x=cur, rec_x=cur_rec,
                   h=buf(Lv, fm[Lv]), out=buf(Lv, fm[Lv]), out_c0=0, drop_p=cfg.dropout[Lv], site=Lv)
        self._res_block_fwd(P, bot, B, img, st, cache_weights, want_out_stats=False)
        blocks.append(bot)
        # ---------------- decoder -------------------------------------------------------------------
        dec_in = bot.out
        ups = []
        for j, i in enumerate(range(Lv - 1, -1, -1)):
            Cup = fm[i + 1]
            Dl, Hl, Wl = S[i + 1]
            wk, bk = f"up_paths.0.{j}.up.weight", f"up_paths.0.{j}.up.bias"
            szl = cfg.z_down[i]
            wp = self._pack(P[wk], L.PK_CT if szl == 2 else L.PK_CT4, Cup, Cup, cache_weights)
            utiles = lib.bpx_convT3d_stats_tiles(Dl, Hl, Wl, szl)
            upart = _Stats.alloc(B, utiles, Cup, dev)
            L.check(lib.bpx_convT3d_k2s2_fwd(self.dt, B, Dl, Hl, Wl, szl, L.tview(dec_in), wp.data_ptr(), P[bk].data_ptr(),
                                             L.tview(cat[i], 0, Cup), upart.data_ptr(), st))
            Ccat = Cup + fm[i]
            pre = f"up_paths.0.{j}.conv_block"
            vox = S[i][0] * S[i][1] * S[i][2]
            rec = _recs(B, Ccat, dev)
            g0, be0 = P[f"{pre}.block.0.weight"], P[f"{pre}.block.0.bias"]
            spart, stiles = out_stats[i]
            if cfg.gn_groups:
                _Stats.finalize_cat([(upart, utiles, Cup), (spart, stiles, fm[i])], B, vox, g0, be0, rec, cfg.gn_groups, st)
            else:
                _Stats.finalize(upart, B, utiles, Cup, vox, g0[:Cup], be0[:Cup], rec, Ccat, 0, st)
                _Stats.finalize(spart, B, stiles, fm[i], vox, g0[Cup:], be0[Cup:], rec, Ccat, Cup, st)
            blk = _Blk(keys=block_keys(pre, False), first=False, S=S[i], cin=Ccat, cout=fm[i], x=cat[i], x_c0=0, rec_x=rec,
                       h=buf(i, fm[i]), out=buf(i, fm[i]), out_c0=0, drop_p=cfg.dropout[i], site=Lv + 1 + j)
            self._res_block_fwd(P, blk, B, img, st, cache_weights, want_out_stats=False)
            blocks.append(blk)
            ups.append((wk, bk, dec_in, Cup, S[i + 1], szl))
            dec_in = blk.out
        # ---------------- heads ----------------------------------------------------------------------
        n_out = sum(cfg.out_channels)
        if len(cfg.out_channels) == 1:
            hw, hb = P["heads.0.weight"], P["heads.0.bias"]
        else:
            hw = torch.cat([P[f"heads.{h}.weight"].reshape(-1, fm[0]) for h in range(len(cfg.out_channels))], 0)
            hb = torch.cat([P[f"heads.{h}.bias"] for h in range(len(cfg.out_channels))], 0)
        hw = hw.reshape(n_out, fm[0]).contiguous()
        feat, So = dec_in, (D0, H0, W0)
        if cfg.post_up:
            # super-resolution: ConvTranspose3d(fm0, fm0, k = s = (post_up, 2, 2)) on the decoder output (resunet.py:399-400)
            So = (D0 * cfg.post_up, H0 * 2, W0 * 2)
            wpu = self._pack(P["post_upsampling.weight"], L.PK_CT if cfg.post_up == 2 else L.PK_CT4, fm[0], fm[0], cache_weights)
            feat = torch.empty((B,) + So + (fm[0],), dtype=T, device=dev)
            pupart = _Stats.alloc(B, lib.bpx_convT3d_stats_tiles(D0, H0, W0, cfg.post_up), fm[0], dev)       # statistics unused
            L.check(lib.bpx_convT3d_k2s2_fwd(self.dt, B, D0, H0, W0, cfg.post_up, L.tview(dec_in), wpu.data_ptr(), P["post_upsampling.bias"].data_ptr(),
                                             L.tview(feat), pupart.data_ptr(), st))
        logits = torch.empty((B, n_out) + So, dtype=torch.float32, device=dev)
        vox0 = So[0] * So[1] * So[2]
        wide = None
        if fm[0] in (16, 32):
            L.check(lib.bpx_head_fwd(self.dt, vox0, B, L.tview(feat), hw.data_ptr(), hb.data_ptr(), n_out, head_act, logits.data_ptr(),
                                     n_out * vox0, vox0, st))
        else:
            # wider first level (e.g. FEATURE_MAPS [48, 64, 80, 96] of the reference's Ovarian-Reserve template): the heads' (n_out, fm0) matrix,
            # zero-padded to 16 rows, runs as a 1x1x1 convolution on the pointwise MFMA kernel; the head kernel then picks the real channels of the
            # 16-channel result (an identity matrix) and applies the output activation.  The logits pass through the 16-bit storage type once.
            w16 = torch.zeros((16, fm[0]), dtype=torch.float32, device=dev)
            b16 = torch.zeros((16,), dtype=torch.float32, device=dev)
            w16[:n_out] = hw
            b16[:n_out] = hb
            o16 = torch.empty((B,) + So + (16,), dtype=T, device=dev)
            wp16 = self._pack(w16, L.PK_DENSE, fm[0], 16, False)
            L.check(lib.bpx_conv1x1_fwd(self.dt, B, vox0, L.tview(feat), wp16.data_ptr(), b16.data_ptr(), L.NULL_T, L.NULL_T, None, L.NULL_T, L.tview(o16), st))
            eye = torch.eye(n_out, 16, dtype=torch.float32, device=dev).contiguous()
            zb = torch.zeros((n_out,), dtype=torch.float32, device=dev)
            L.check(lib.bpx_head_fwd(self.dt, vox0, B, L.tview(o16), eye.data_ptr(), zb.data_ptr(), n_out, head_act, logits.data_ptr(), n_out * vox0, vox0, st))
            wide = dict(o16=o16, w16=w16, eye=eye)
        if cfg.ndim == 2:
            logits = logits.reshape(B, n_out, So[1], So[2])
        ctx = None
        if save:
            ctx = dict(B=B, S=S, So=So, img=img, x_ndhwc=x_ndhwc, blocks=blocks, cat=cat, pools=pools, ups=ups, feat=feat, dec_out=dec_in, hw=hw,
                       Pw=(P if P is not P_orig else None), want_dx=want_dx, wide_head=wide)
        return logits, ctx

    # ------------------------------------------------------------------------------------------
    def _block_bwd(self, P, G, blk: _Blk, B, dOut: L.Tensor, img, st, dx_extra: Optional[L.Tensor] = None, dx_out: Optional[L.Tensor] = None):
        """Backward of one residual block.  dOut: gradient of the block output (T, NDHWC view).
        Writes parameter grads into G; writes the input gradient into dx_out (a view with blk.cin channels)."""
        D, H, W = blk.S
        k = blk.keys
        dev = blk.h.device
        C1 = blk.cout
        vox = D * H * W
        T = self.gdtype
        # conv2 weight/bias grad, shortcut weight grad
        # both biases add to the same tensor: identical gradients, written by the same reduction
        dropped = blk.a is not None                      # training-mode dropout: conv2 read the materialised, masked activation blk.a
        fused2 = not dropped and self._bwd_fused_ok(B, blk.S, C1, dOut.C)
        if dropped:
            self._wgrad(B, blk.S, L.tview(blk.a), None, 0, dOut, 3, G[k["w2"]], G[k["b2"]], st, dev, db2=G[k["bsc"]])
        elif not fused2:
            self._wgrad(B, blk.S, L.tview(blk.h), blk.rec_h, self.act, dOut, 3, G[k["w2"]], G[k["b2"]], st, dev, db2=G[k["bsc"]])
        # decoder blocks at the large levels (round 6): the shortcut's weight gradient rides along in the pass that forms the block's input gradient
        # (bpx_conv1x1_fwd_split_wgrad below streams both of its operands anyway); everywhere else it is a launch of its own
        sc_ws = 0
        if (isinstance(dx_out, tuple) and blk.rec_x is not None and self._deferred and not self.use_side_stream and self.cfg.gn_groups == 0
                and dOut.C * 3 == blk.cin and self.dtype != torch.float32):
            sc_ws = int(lib.bpx_conv1x1_fwd_split_wgrad_workspace(self.bdt, B, vox, dOut.C))
        if blk.first and self.cfg.in_ch == 1:
            if getattr(self, "_r1_done", False):       # formed by bpx_maxpool3d_bwd_r1, the pass that wrote dOut (see _backward)
                self._r1_done = False
            else:
                ws1 = self._workspace(lib.bpx_conv1x1_c1_wgrad_workspace(C1), dev)
                self._run_side(dev, lambda s_: L.check(lib.bpx_conv1x1_c1_wgrad(self.gdt, B * vox, img.data_ptr(), dOut, G[k["wsc"]].data_ptr(),
                                                                                ws1.data_ptr(), ws1.numel(), s_)))
        elif not sc_ws:
            self._wgrad(B, blk.S, L.tview(blk.x, blk.x_c0, blk.cin), None, 0, dOut, 1, G[k["wsc"]], None, st, dev)
        # conv2 dgrad fused with ELU' and the InstanceNorm reductions
        g1 = torch.empty((B, D, H, W, C1), dtype=T, device=dev)
        self._keep.append(g1)   # read by the side-stream wgrad of conv1
        w2t = self._pack(P[k["w2"]], L.PK_K3_T, C1, C1, False)
        if dropped:   # plain dgrad -> gradient of the masked activation; then mask, activation derivative and the IN-backward sums in one pass
            L.check(lib.bpx_conv3d_dgrad(self.gdt, B, D, H, W, dOut, w2t.data_ptr(), L.NULL_T, None, 0, L.tview(g1), None, st))
            tiles = lib.bpx_norm_act_dropout_tiles(self.gdt, vox, C1)
            red = torch.empty((B, tiles, 2, C1), dtype=torch.float32, device=dev)
            mptr, mmode = self._drop_mask(blk, g1.numel(), dev)
            L.check(lib.bpx_norm_act_dropout_bwd(self.bdt, B, vox, L.tview(g1), L.tview(blk.h), blk.rec_h.data_ptr(), self.act, blk.drop_p, self.drop_seed,
                                                 blk.drop_ctr.data_ptr(), blk.site, mptr, 1 if mmode else 0, L.tview(g1), red.data_ptr(), st))
        elif fused2:   # dgrad + wgrad of conv2 in one pass over (dOut, h)
            tiles, red = self._bwd_fused(B, blk.S, dOut, w2t, L.tview(blk.h), blk.rec_h, L.tview(g1), G[k["w2"]], G[k["b2"]], G[k["bsc"]], st, dev)
        else:
            tiles = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, C1)
            red = torch.empty((B, tiles, 2, C1), dtype=torch.float32, device=dev)
            L.check(lib.bpx_conv3d_dgrad(self.bdt, B, D, H, W, dOut, w2t.data_ptr(), L.tview(blk.h), blk.rec_h.data_ptr(), self.act,
                                         L.tview(g1), red.data_ptr(), st))
        coef = torch.empty((B, C1, 4), dtype=torch.float32, device=dev)
        # (deferred form: dgamma / dbeta arrive with the flush of the weight-gradient reductions; `red` holds their per-sample terms until then)
        self._keep.append(red)
        L.check(_NBF(red.data_ptr(), B, tiles, C1, vox, blk.rec_h.data_ptr(), P[k["g1"]].data_ptr(),
                                                   G[k["g1"]].data_ptr(), G[k["be1"]].data_ptr(), self.cfg.gn_groups or C1, coef.data_ptr(), st))
        first_c1 = blk.first and self.cfg.in_ch == 1
        if first_c1 and lib.bpx_conv3d_c1_wgrad_nb_supported(self.bdt, W) and os.environ.get("BPX_C1_NB", "1") != "0":
            # the first layer has no input gradient: its weight gradient is the only reader of dH = a * g1 + b * h + c0, which is therefore formed
            # inside that kernel and never stored (bpx_norm_bwd_apply's pass over three tensor units is gone)
            wsc = self._workspace(lib.bpx_conv3d_c1_wgrad_workspace(C1), dev)
            self._keep.append(coef)
            self._run_side(dev, lambda s_: L.check(lib.bpx_conv3d_c1_wgrad_nb(self.bdt, B, D, H, W, img.data_ptr(), L.tview(g1), L.tview(blk.h), coef.data_ptr(),
                                                                              G[k["w1"]].data_ptr(), G[k["b1"]].data_ptr(), wsc.data_ptr(), wsc.numel(), s_)))
            self._keep.append(g1)
            return
        L.check(lib.bpx_norm_bwd_apply(self.bdt, B, vox, L.tview(g1), L.tview(blk.h), coef.data_ptr(), L.NULL_T, L.tview(g1), st))
        dH = L.tview(g1)
        # conv1
        if first_c1:
            wsc = self._workspace(lib.bpx_conv3d_c1_wgrad_workspace(C1), dev)
            self._run_side(dev, lambda s_: L.check(lib.bpx_conv3d_c1_wgrad(self.gdt, B, D, H, W, img.data_ptr(), dH, G[k["w1"]].data_ptr(),
                                                                           G[k["b1"]].data_ptr(), wsc.data_ptr(), wsc.numel(), s_)))
            self._keep.append(g1)
            return
        xv = L.tview(blk.x, blk.x_c0, blk.cin)
        has_norm = blk.rec_x is not None
        Cx = blk.cin
        fused1 = has_norm and dx_out is not None and self._bwd_fused_ok(B, blk.S, Cx, C1)
        if not fused1:
            self._wgrad(B, blk.S, xv, blk.rec_x, self.act if has_norm else 0, dH, 3, G[k["w1"]], G[k["b1"]], st, dev)
        if dx_out is None:
            return
        g0 = torch.empty((B, D, H, W, Cx), dtype=T, device=dev)
        tiles0 = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, Cx)
        w1t = self._pack(P[k["w1"]], L.PK_K3_T, Cx, C1, False)
        wsct = self._pack(P[k["wsc"]], L.PK_DENSE_T, Cx, C1, False)
        if has_norm:
            if fused1:   # dgrad + wgrad of conv1 in one pass over (dH, x)
                tiles0, red0 = self._bwd_fused(B, blk.S, dH, w1t, xv, blk.rec_x, L.tview(g0), G[k["w1"]], G[k["b1"]], None, st, dev)
            else:
                red0 = torch.empty((B, tiles0, 2, Cx), dtype=torch.float32, device=dev)
                L.check(lib.bpx_conv3d_dgrad(self.bdt, B, D, H, W, dH, w1t.data_ptr(), xv, blk.rec_x.data_ptr(), self.act, L.tview(g0),
                                             red0.data_ptr(), st))
            coef0 = torch.empty((B, Cx, 4), dtype=torch.float32, device=dev)
            gng = self.cfg.gn_groups
            if gng and (Cx // gng) not in (1, 2, 4, 8, 16, 32, 64):
                # the concatenated decoder input: 6 / 12 / 24 / 48 channels per group -> per-channel totals, then the general group kernel
                sums0 = torch.empty((B, Cx, 2), dtype=torch.float64, device=dev)
                L.check(lib.bpx_norm_channel_sums(red0.data_ptr(), B, tiles0, Cx, sums0.data_ptr(), Cx, 0, st))
                L.check(lib.bpx_groupnorm_bwd_finalize(sums0.data_ptr(), B, Cx, vox, blk.rec_x.data_ptr(), P[k["g0"]].data_ptr(), G[k["g0"]].data_ptr(),
                                                       G[k["be0"]].data_ptr(), gng, coef0.data_ptr(), st))
            else:
                self._keep.append(red0)
                L.check(_NBF(red0.data_ptr(), B, tiles0, Cx, vox, blk.rec_x.data_ptr(), P[k["g0"]].data_ptr(),
                                                           G[k["g0"]].data_ptr(), G[k["be0"]].data_ptr(), gng or Cx, coef0.data_ptr(), st))
            if isinstance(dx_out, tuple):   # decoder block: the gradient of the concatenated input leaves as its (up, skip) parts
                assert dx_extra is None
                if sc_ws:
                    wsw = self._workspace(sc_ws, dev)
                    L.check(lib.bpx_conv1x1_fwd_split_wgrad(self.bdt, B, vox, dOut, wsct.data_ptr(), L.tview(g0), xv, coef0.data_ptr(), dx_out[0], dx_out[1],
                                                            G[k["wsc"]].data_ptr(), wsw.data_ptr(), wsw.numel(), st))
                else:
                    L.check(lib.bpx_conv1x1_fwd_split(self.bdt, B, vox, dOut, wsct.data_ptr(), None, L.tview(g0), xv, coef0.data_ptr(),
                                                      L.NULL_T, dx_out[0], dx_out[1], st))
            else:
                L.check(lib.bpx_conv1x1_fwd(self.bdt, B, vox, dOut, wsct.data_ptr(), None, L.tview(g0), xv, coef0.data_ptr(),
                                            dx_extra if dx_extra is not None else L.NULL_T, dx_out, st))
        else:
            L.check(lib.bpx_conv3d_dgrad(self.gdt, B, D, H, W, dH, w1t.data_ptr(), L.NULL_T, None, 0, L.tview(g0), None, st))
            L.check(lib.bpx_conv1x1_fwd(self.gdt, B, vox, dOut, wsct.data_ptr(), None, L.NULL_T, L.NULL_T, None, L.tview(g0), dx_out, st))

    def backward(self, P: Dict[str, torch.Tensor], ctx, dlogits: torch.Tensor, on_last_block=None) -> Dict[str, torch.Tensor]:
        """``on_last_block``: called (no arguments) right before the backward of the FIRST encoder block - the last stretch of the pass.  At that
        point every queued weight-gradient reduction has been flushed, so all parameter gradients except ``down_path.0.*`` are final in the flat
        slab (``self.last_flat_grad``, parameter order): a data-parallel step starts their all-reduce there and lets it run beside the rest of
        the backward (graphs.DataParallelTrainStep; what DDP's buckets do for the reference, base_workflow.py:952-958)."""
        self._on_last_block = on_last_block
        self._keep = []   # buffers the side stream may still be reading; released after the final stream join
        # the ~29 weight-gradient reductions of a step run as one batched launch at the end (they are latency chains of a
        # few hundred blocks each; back to back they cost 0.6 ms).  Not with the side stream: the flush is stream-ordered.
        self._deferred = not self.use_side_stream      # (the small-level side branch is joined at the end of _backward, before the flush)
        self._side_used = False
        self._after_flush = []
        if self._deferred:
            L.check(lib.bpx_wgrad_defer_begin())
        Pw = ctx.get("Pw")
        try:
            G = self._backward(P if Pw is None else Pw, ctx, dlogits)
        finally:
            self._on_last_block = None
            if self._deferred:
                self._deferred = False
                L.check(lib.bpx_wgrad_defer_flush(L.stream_ptr()))
            self._keep = []
        for fn in self._after_flush:
            fn()
        self._after_flush = []
        if Pw is None:
            return G
        if self._pad_plan is not None:                        # zero-padded widths: the parameters' own rows / columns of the padded gradients
            G = unpad_channel_grads(G, self._pad_plan)
        return unlift_grads(G, P)                            # after the flush: it is the flush that writes the conv gradients

    def _backward(self, P: Dict[str, torch.Tensor], ctx, dlogits: torch.Tensor) -> Dict[str, torch.Tensor]:
        cfg = self.cfg
        B, S, img = ctx["B"], ctx["S"], ctx["img"]
        blocks: List[_Blk] = ctx["blocks"]
        cat, pools, ups, feat = ctx["cat"], ctx["pools"], ctx["ups"], ctx["feat"]
        fm = list(cfg.feature_maps)
        Lv = cfg.depth
        dev = dlogits.device
        st = L.stream_ptr()
        T = self.gdtype
        names = list(P.keys())
        sizes = [P[n].numel() for n in names]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.last_flat_grad = flat
        G: Dict[str, torch.Tensor] = {}
        o = 0
        for n, s in zip(names, sizes):
            G[n] = flat[o:o + s].view(P[n].shape)
            o += s
        # ---- head -------------------------------------------------------------------------------
        n_out = sum(cfg.out_channels)
        D0, H0, W0 = S[0]
        So = ctx.get("So", S[0])
        vox0 = So[0] * So[1] * So[2]
        dl = dlogits.contiguous().float()
        dfeat = torch.empty((B,) + tuple(So) + (fm[0],), dtype=T, device=dev)
        one_head = len(cfg.out_channels) == 1                     # its gradients are written in place (G is zeroed); several heads: split below
        hwg = G["heads.0.weight"] if one_head else torch.zeros((n_out, fm[0]), dtype=torch.float32, device=dev)
        hbg = G["heads.0.bias"] if one_head else torch.zeros((n_out,), dtype=torch.float32, device=dev)
        wide = ctx.get("wide_head")
        if wide is None:
            hws = self._workspace(lib.bpx_head_bwd_workspace(fm[0], n_out), dev)
            L.check(lib.bpx_head_bwd(self.bdt, vox0, B, L.tview(feat), ctx["hw"].data_ptr(), n_out, dl.data_ptr(), n_out * vox0, vox0,
                                     L.tview(dfeat), hwg.data_ptr(), hbg.data_ptr(), hws.data_ptr(), hws.numel(), st))
        else:
            # wide head (forward above): the head kernel's backward on the 16-channel tensor gives its gradient (the identity pick's own gradients are
            # discarded), then the 1x1x1 convolution's two gradients: k = 1 wgrad (rows 0 .. n_out of the padded matrix are the heads') and the
            # pointwise GEMM with the transposed matrix.  The weight gradient is reduced right away (not with the step's batch): it is copied below.
            do16 = torch.empty((B,) + tuple(So) + (16,), dtype=T, device=dev)
            eg, ebg = torch.zeros((n_out, 16), dtype=torch.float32, device=dev), torch.zeros((n_out,), dtype=torch.float32, device=dev)
            hws = self._workspace(lib.bpx_head_bwd_workspace(16, n_out), dev)
            L.check(lib.bpx_head_bwd(self.bdt, vox0, B, L.tview(wide["o16"]), wide["eye"].data_ptr(), n_out, dl.data_ptr(), n_out * vox0, vox0,
                                     L.tview(do16), eg.data_ptr(), ebg.data_ptr(), hws.data_ptr(), hws.numel(), st))
            dw16 = torch.zeros((16, fm[0], 1, 1, 1), dtype=torch.float32, device=dev)
            db16 = torch.zeros((16,), dtype=torch.float32, device=dev)
            ws16 = self._workspace(lib.bpx_conv3d_wgrad_workspace(B, So[0], So[1], So[2], fm[0], 16, 1), dev)
            L.check(lib.bpx_conv3d_wgrad_db2(self.bdt, B, So[0], So[1], So[2], L.tview(feat), None, 0, L.tview(do16), 1, dw16.data_ptr(), db16.data_ptr(), None,
                                             ws16.data_ptr(), ws16.numel(), st))         # on this stream (not the optional side stream): read right below
            if self._deferred:
                L.check(lib.bpx_wgrad_defer_flush(L.stream_ptr()))
                L.check(lib.bpx_wgrad_defer_begin())
            hwg.view(n_out, fm[0]).copy_(dw16.view(16, fm[0])[:n_out])
            hbg.copy_(db16[:n_out])
            wt16 = self._pack(wide["w16"], L.PK_DENSE_T, fm[0], 16, False)
            L.check(lib.bpx_conv1x1_fwd(self.gdt, B, vox0, L.tview(do16), wt16.data_ptr(), None, L.NULL_T, L.NULL_T, None, L.NULL_T, L.tview(dfeat), st))
            self._keep += [do16, dw16, db16, eg, ebg]
        if cfg.post_up:
            dec_out, dup_feat = ctx["dec_out"], dfeat
            wsn = lib.bpx_convT3d_k2s2_wgrad_workspace(B, D0, H0, W0, cfg.post_up, fm[0], fm[0])
            ws = self._workspace(wsn, dev)
            L.check(lib.bpx_convT3d_k2s2_wgrad(self.bdt, B, D0, H0, W0, cfg.post_up, L.tview(dec_out), L.tview(dup_feat), G["post_upsampling.weight"].data_ptr(),
                                               G["post_upsampling.bias"].data_ptr(), ws.data_ptr(), ws.numel(), st))
            dfeat = torch.empty((B, D0, H0, W0, fm[0]), dtype=T, device=dev)
            wt = self._pack(P["post_upsampling.weight"], L.PK_CT_T if cfg.post_up == 2 else L.PK_CT4_T, fm[0], fm[0], False)
            L.check(lib.bpx_convT3d_k2s2_dgrad(self.gdt, B, D0, H0, W0, cfg.post_up, L.tview(dup_feat), wt.data_ptr(), L.tview(dfeat), st))
            self._keep.append(dup_feat)
        o = 0
        for h, oc in enumerate(cfg.out_channels if not one_head else ()):
            G[f"heads.{h}.weight"].copy_(hwg[o:o + oc].view(G[f"heads.{h}.weight"].shape))
            G[f"heads.{h}.bias"].copy_(hbg[o:o + oc])
            o += oc
        # ---- decoder (blocks list: enc 0..Lv-1, bottleneck, dec j=0..Lv-1 for levels Lv-1..0) -------
        dskip: List[Optional[torch.Tensor]] = [None] * Lv      # d(concat) leaves the decoder block as two dense tensors
        dOut = L.tview(dfeat)
        keep = [dfeat]
        for j in range(Lv - 1, -1, -1):       # decoder block j handles level i = Lv-1-j; walk from level 0 up
            i = Lv - 1 - j
            blk = blocks[Lv + 1 + j]
            wk, bk, x_in, Cup, Sl, szl = ups[j]
            dup = torch.empty((B,) + S[i] + (Cup,), dtype=T, device=dev)
            dskip[i] = torch.empty((B,) + S[i] + (fm[i],), dtype=T, device=dev)
            self._block_bwd(P, G, blk, B, dOut, img, st, None, (L.tview(dup), L.tview(dskip[i])))
            # transposed conv backward
            keep.append(dup)
            dUp = L.tview(dup)
            wsn = lib.bpx_convT3d_k2s2_wgrad_workspace(B, Sl[0], Sl[1], Sl[2], szl, Cup, Cup)
            ws = self._workspace(wsn, dev)
            self._run_side(dev, lambda s_, x_in=x_in, dUp=dUp, wk=wk, bk=bk, Sl=Sl, ws=ws, szl=szl: L.check(lib.bpx_convT3d_k2s2_wgrad(
                self.bdt, B, Sl[0], Sl[1], Sl[2], szl, L.tview(x_in), dUp, G[wk].data_ptr(), G[bk].data_ptr(), ws.data_ptr(), ws.numel(), s_)), Sl)
            dxin = torch.empty((B,) + Sl + (Cup,), dtype=T, device=dev)
            wt = self._pack(P[wk], L.PK_CT_T if szl == 2 else L.PK_CT4_T, Cup, Cup, False)
            L.check(lib.bpx_convT3d_k2s2_dgrad(self.gdt, B, Sl[0], Sl[1], Sl[2], szl, dUp, wt.data_ptr(), L.tview(dxin), st))
            dOut = L.tview(dxin)
            keep.append(dxin)
        # ---- bottleneck -------------------------------------------------------------------------
        dP = torch.empty((B,) + S[Lv] + (fm[Lv - 1],), dtype=T, device=dev)
        self._block_bwd(P, G, blocks[Lv], B, dOut, img, st, None, L.tview(dP))
        # ---- encoder ------------------------------------------------------------------------------
        for i in range(Lv - 1, -1, -1):
            D, H, W = S[i]
            Cup = fm[i + 1]
            # dOut_i = dSkip + unpool(dP); written in place over dSkip
            skipv = L.tview(dskip[i])
            r1_ws = 0
            if i == 0 and cfg.in_ch == 1 and self._deferred and not self.use_side_stream and self.dtype != torch.float32:
                # level 0 of a one-channel-image network (round 6): the first block's rank-1 shortcut weight gradient is a sum over the tensor this pass writes
                r1_ws = int(lib.bpx_maxpool3d_bwd_r1_workspace(self.bdt, B, D, H, W, cfg.z_down[i], fm[i]))
            if r1_ws:
                wsr = self._workspace(r1_ws, dev)
                L.check(lib.bpx_maxpool3d_bwd_r1(self.bdt, B, D, H, W, cfg.z_down[i], L.tview(cat[i], Cup, fm[i]), L.tview(dP), skipv, skipv, img.data_ptr(),
                                                 G[blocks[0].keys["wsc"]].data_ptr(), wsr.data_ptr(), wsr.numel(), st))
                self._r1_done = True
            else:
                L.check(lib.bpx_maxpool3d_bwd(self.bdt, B, D, H, W, cfg.z_down[i], L.tview(cat[i], Cup, fm[i]), L.tview(dP), skipv, skipv, st))
            if i == 0 and getattr(self, "_on_last_block", None) is not None:
                if self._deferred:      # the reductions queued so far write their gradients now; the last block's are queued afresh
                    L.check(lib.bpx_wgrad_defer_flush(L.stream_ptr()))
                    L.check(lib.bpx_wgrad_defer_begin())
                self._on_last_block()
            if i > 0:
                dPn = torch.empty((B,) + S[i] + (fm[i - 1],), dtype=T, device=dev)
                self._block_bwd(P, G, blocks[i], B, skipv, img, st, None, L.tview(dPn))
                keep.append(dP)
                dP = dPn
            elif ctx.get("want_dx"):
                dx0 = torch.empty((B,) + S[0] + (cfg.in_ch,), dtype=T, device=dev)
                self._block_bwd(P, G, blocks[0], B, skipv, img, st, None, L.tview(dx0))
                G["__dx__"] = dx0
            else:
                self._block_bwd(P, G, blocks[0], B, skipv, img, st, None, None)  # the image needs no gradient
        if self._side_used and self._side_stream is not None:
            torch.cuda.current_stream(dev).wait_stream(self._side_stream)
        return G
