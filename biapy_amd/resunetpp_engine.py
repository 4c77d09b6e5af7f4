"""Executor of ResUNet++ (3D) on the MI355X kernels (cfg 4: instance segmentation with B / C / D channels).

Host side of ``biapy/models/resunet++.py:412-466`` (forward graph) and of its autograd graph, built from the C-ABI kernels of
include/biapy_amd.h.  Every tensor is NDHWC in the storage dtype (bf16, or f32 = exact mode); PyTorch only owns the memory.

What the graph needs beyond the plain ResUNet (engine.py) and how it runs here:

  * residual block with a 3x3x3 shortcut + norm (``ResConvBlock(skip_k_size=3, skip_norm=...)``, blocks.py:1366-1378, :1456-1459):
    ``out = conv2(ELU(IN(conv1(ELU(IN(x)))))) + IN(conv_s(x))``.  The two InstanceNorm+ELU inside the main branch are prologues of
    the consuming convolutions exactly as in engine.py (never materialised); the shortcut is a third convolution of the raw input
    and the sum ``main + scale_s * s + shift_s`` is one streaming pass (``bpx_channel_affine``);
  * squeeze-and-excitation (blocks.py:1119-1191): channel means from ``bpx_tensor_stats``, the two bias-free Linear layers on the
    (B, C) vector are PyTorch device ops (a few hundred FLOPs; backward written out), the recalibration is ``bpx_channel_affine``,
    its gradient needs ``bpx_dot_stats``;
  * ASPP (heads.py:13-133): a 3x3x3 convolution with dilation d is d^3 ordinary convolutions on the sub-lattices x = r (mod d)
    (dilation.py: ``bpx_gather3d_tables`` into the batch dimension, the ordinary conv / dgrad / wgrad kernels, ``bpx_scatter3d_tables``
    back); conv -> ReLU -> IN is materialised (``bpx_norm_act_fwd`` twice around ``bpx_tensor_stats``) into channel slices of the
    concatenation buffer; the closing 1x1x1 convolution is ``bpx_conv1x1_fwd``;
  * attention gate (blocks.py:2168-2298): IN -> ReLU -> conv3 on both inputs (prologue form), max-pool of the encoder branch, sum,
    IN -> ReLU (materialised) -> 1x1x1 conv to ONE channel (run zero-padded to 16 outputs on ``bpx_conv1x1_fwd``), and
    ``bpx_gate_mul_fwd`` for ``gate * x2``;
  * ``ResUpBlock`` (blocks.py:1603-1655): ``bpx_convT3d_k2s2_fwd`` into channels [0, Cup) of the level's concatenation buffer, the
    bridge copied into [Cup, Cup + Cskip) (``bpx_norm_act_fwd`` with an identity record).

The backward is a tape: every forward step appends the closure that turns the gradient of its output into gradients of its
inputs and parameters, and ``backward`` runs the tape in reverse.  A tensor with several consumers (encoder features: next block,
attention, skip connection; decoder tensors: attention branch and gate) accumulates its gradient.
Correctness-first (row X did not exist before round 2): pinned to the reference's own outputs (tests/golden/resunetpp_golden.npz),
not tuned - the dilated convolutions in particular run on many small sub-lattice volumes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import os

import torch

from . import _lib as L
from . import dilation
from .engine import NetConfig, ResUNetEngine, _recs, _Stats

lib = L.lib


@dataclass
class PPConfig:
    in_ch: int
    feature_maps: Sequence[int]
    out_channels: Sequence[int] = (1,)
    activation: str = "elu"
    z_down: Optional[Sequence[int]] = None      # one entry per pooled level (levels 1..depth), each 1 or 2
    rates: Sequence[int] = (6, 12, 18)

    def __post_init__(self):
        fm = list(self.feature_maps)
        if len(fm) < 3:
            raise ValueError("ResUNet++ needs at least three feature maps (depth = len - 2)")
        self.depth = len(fm) - 2
        zd = [2] * (self.depth + 1) if self.z_down is None else [int(v) for v in list(self.z_down)[: self.depth + 1]]
        if len(zd) != self.depth + 1 or any(v not in (1, 2) for v in zd):
            raise NotImplementedError(f"z_down={self.z_down!r}: one value per level, each 1 or 2")
        self.z_down = tuple(zd)
        if self.in_ch != 1:
            raise NotImplementedError("ResUNet++ engine: one input channel (the first-layer kernels are the Cin = 1 ones)")
        if any(c % 16 for c in fm):
            raise NotImplementedError(f"feature_maps {fm} must be multiples of 16 (MFMA tile)")
        if fm[0] not in (16, 32) or sum(self.out_channels) > 4:
            raise NotImplementedError("output head supports <= 4 channels from 16 or 32 features")
        if self.activation not in L.ACT:
            raise NotImplementedError(f"activation={self.activation!r} is not implemented on the MI355X engine")


class _V:
    """A dense NDHWC tensor on the tape + its spatial extent + its gradient (a dense tensor of the same shape, or None)."""

    def __init__(self, buf: torch.Tensor, S: Tuple[int, int, int]):
        self.buf, self.S, self.C = buf, tuple(S), buf.shape[-1]
        self.grad: Optional[torch.Tensor] = None
        self.grad_shared = False        # grad is ALSO another tensor's gradient (read-only): anything that adds to it in place copies it first

    def view(self) -> "L.Tensor":
        return L.tview(self.buf)

    @property
    def vox(self) -> int:
        return self.S[0] * self.S[1] * self.S[2]


@dataclass
class _Nrm:
    """InstanceNorm + activation applied as the PROLOGUE of a consuming convolution."""
    rec: torch.Tensor          # [B][C]{mean, rstd, scale, shift}
    act: int
    gamma: torch.Tensor
    dgamma: Optional[torch.Tensor]
    dbeta: Optional[torch.Tensor]


class ResUNetPPEngine(ResUNetEngine):
    def __init__(self, cfg: PPConfig, dtype: torch.dtype = torch.bfloat16):
        # the base class provides weight packing, the wgrad workspace / deferred reduction and the stream plumbing
        super().__init__(NetConfig(in_ch=1, feature_maps=[cfg.feature_maps[0], cfg.feature_maps[1]], out_channels=tuple(cfg.out_channels),
                                   activation=cfg.activation), dtype)
        self.pp = cfg
        self.relu = L.ACT["relu"]
        self._const: Dict = {}

    # ------------------------------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------------------------------
    def _new(self, S, C) -> _V:
        return _V(torch.empty((self._B,) + tuple(S) + (C,), dtype=self.dtype, device=self._dev), S)

    def _c(self, key, make):
        k = (key, str(self._dev))
        if k not in self._const:
            self._const[k] = make()
        return self._const[k]

    def _ident_rec(self, C):
        def make():
            rec = torch.zeros((self._B, C, 4), dtype=torch.float32, device=self._dev)
            rec[..., 1] = 1.0
            rec[..., 2] = 1.0                                     # mean 0, rstd 1, scale 1, shift 0: the pass is the activation alone
            return rec
        return self._c(("ident", self._B, C), make)

    def _ones_bc(self, C):
        return self._c(("ones_bc", self._B, C), lambda: torch.ones((self._B, C), dtype=torch.float32, device=self._dev))

    def _accum(self, v: _V, g: torch.Tensor, c0: int = 0) -> None:
        """v.grad += g[..., c0:c0+v.C].  A dense first contribution is adopted (the caller gives up ownership)."""
        dense = g.shape[-1] == v.C and c0 == 0
        if v.grad is None and dense:
            v.grad = g
            return
        gv = L.tview(g, c0, v.C)
        if v.grad is None:
            v.grad = torch.empty((self._B,) + v.S + (v.C,), dtype=self.gdtype, device=self._dev)
            L.check(lib.bpx_norm_act_fwd(self.gdt, self._B, v.vox, gv, self._ident_rec(v.C).data_ptr(), 0, L.tview(v.grad), self._st))   # copy of a slice (gradient tensors only)
        else:
            if v.grad_shared:
                v.grad, v.grad_shared = v.grad.clone(), False
            L.check(lib.bpx_channel_affine(self.gdt, self._B, v.vox, L.tview(v.grad), gv, self._ones_bc(v.C).data_ptr(), None, L.tview(v.grad), self._st))
        self._keep.append(g)

    def _acc_target(self, v: _V):
        """For a kernel that can ADD to an existing tensor while it writes its result (norm_bwd_apply, channel_affine, conv1x1): the
        (addend view, destination tensor, fresh?) that make it leave ``v.grad (+)= result`` - the gradient so far as the addend and the
        destination, or a new tensor the caller adopts as ``v.grad``.  Saves the separate accumulation pass of ``_accum``."""
        if v.grad is None:
            return L.NULL_T, torch.empty(v.buf.shape, dtype=self.gdtype, device=self._dev), True
        if v.grad_shared:
            v.grad, v.grad_shared = v.grad.clone(), False
        return L.tview(v.grad), v.grad, False

    def _tensor_part(self, v: _V):
        tiles = lib.bpx_tensor_stats_tiles(v.vox)
        part = _Stats.alloc(self._B, tiles, v.C, self._dev)
        L.check(lib.bpx_tensor_stats(self.dt, self._B, v.vox, v.view(), part.data_ptr(), self._st))
        return part, tiles

    def _finalize(self, part, tiles, C, vox, gamma, beta, rec=None, rec_ld=None, rec_off=0):
        if rec is None:
            rec = _recs(self._B, C, self._dev)
        _Stats.finalize(part, self._B, tiles, C, vox, gamma, beta, rec, C if rec_ld is None else rec_ld, rec_off, self._st)
        return rec

    def _stats_rec(self, v: _V, gamma, beta):
        part, tiles = self._tensor_part(v)
        return self._finalize(part, tiles, v.C, v.vox, gamma, beta)

    def _in_bwd(self, raw: _V, rec, act, dA: torch.Tensor, gamma, dgamma, dbeta) -> torch.Tensor:
        """Backward of a materialised ``act(IN(raw))``: dA (dense) -> d(raw)."""
        B, C, vox = self._B, raw.C, raw.vox
        # (mixed mode: raw is the forward pass's fp16 tensor, every gradient tensor bf16 - engine.ResUNetEngine's codes: gdt = kernels on gradient
        #  tensors only, bdt = backward kernels that also read a forward tensor)
        tiles = lib.bpx_norm_act_tiles(self.gdt, vox, C)
        red = torch.empty((B, tiles, 2, C), dtype=torch.float32, device=self._dev)
        g = torch.empty((B,) + raw.S + (C,), dtype=self.gdtype, device=self._dev)
        L.check(lib.bpx_norm_act_bwd(self.bdt, B, vox, L.tview(dA), raw.view(), rec.data_ptr(), act, L.NULL_T, L.tview(g), red.data_ptr(), self._st))
        coef = torch.empty((B, C, 4), dtype=torch.float32, device=self._dev)
        L.check(lib.bpx_norm_bwd_finalize(red.data_ptr(), B, tiles, C, vox, rec.data_ptr(), gamma.data_ptr(), L.ptr(dgamma), L.ptr(dbeta), C,
                                          coef.data_ptr(), self._st))
        L.check(lib.bpx_norm_bwd_apply(self.bdt, B, vox, L.tview(g), raw.view(), coef.data_ptr(), L.NULL_T, L.tview(g), self._st))
        self._keep.append(dA)
        return g

    def _act_bwd(self, raw: _V, act, dA: torch.Tensor) -> torch.Tensor:
        """Backward of a materialised activation alone: dA * act'(raw)."""
        B, C, vox = self._B, raw.C, raw.vox
        red = torch.empty((B, lib.bpx_norm_act_tiles(self.gdt, vox, C), 2, C), dtype=torch.float32, device=self._dev)
        g = torch.empty((B,) + raw.S + (C,), dtype=self.gdtype, device=self._dev)
        L.check(lib.bpx_norm_act_bwd(self.bdt, B, vox, L.tview(dA), raw.view(), self._ident_rec(C).data_ptr(), act, L.NULL_T, L.tview(g), red.data_ptr(), self._st))
        self._keep += [dA, red]
        return g

    # ------------------------------------------------------------------------------------------------------------------
    # tape operations: forward now, the backward closure goes onto self._tape (when training)
    # ------------------------------------------------------------------------------------------------------------------
    def _conv3(self, x: _V, wk: str, bk: str, Cout: int, nrm: Optional[_Nrm] = None, want_stats: bool = True, batch: Optional[int] = None):
        """y = conv3x3x3(act(IN(x))) with the normalisation + activation as the conv's prologue (nrm given), or conv3x3x3(x)."""
        P, G, B = self._P, self._G, (self._B if batch is None else batch)
        D, H, W = x.S
        y = _V(torch.empty((B, D, H, W, Cout), dtype=self.dtype, device=self._dev), x.S)
        tiles = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, Cout)
        part = _Stats.alloc(B, tiles, Cout, self._dev) if want_stats else None
        wp = self._pack(P[wk], L.PK_K3, x.C, Cout, False)
        L.check(lib.bpx_conv3d_fwd(self.dt, B, D, H, W, x.view(), L.ptr(nrm.rec) if nrm else None, nrm.act if nrm else 0, wp.data_ptr(), P[bk].data_ptr(),
                                   L.NULL_T, None, None, y.view(), L.ptr(part), self._st))
        if G is not None:
            def bwd():
                dy = L.tview(y.grad)
                fused = (nrm is not None and nrm.act <= 3 and os.environ.get("BPX_BWD_FUSED", "1") != "0" and self.dtype != torch.float32 and not self.use_side_stream
                         and x.view().cs == 0 and bool(lib.bpx_conv3d_bwd_fused_supported(self.bdt, B, D, H, W, x.C, Cout)))
                if not fused:
                    self._wgrad(B, x.S, x.view(), nrm.rec if nrm else None, nrm.act if nrm else 0, dy, 3, G[wk], G[bk], self._st, self._dev)
                wt = self._pack(P[wk], L.PK_K3_T, x.C, Cout, False)
                g = torch.empty((B, D, H, W, x.C), dtype=self.gdtype, device=self._dev)
                if nrm is not None and fused:
                    # round 5 (first part): dgrad (+ act', + IN-backward sums) and wgrad (+ bias gradient) of the conv in ONE pass
                    # over (dy, x) - the level-0 / level-1 shapes the cfg-2 engine fuses too (bpx_conv3d_bwd_fused_supported)
                    rt = lib.bpx_conv3d_bwd_fused_stats_tiles(B, D, H, W, x.C, Cout)
                    red = torch.empty((B, rt, 2, x.C), dtype=torch.float32, device=self._dev)
                    ws = self._workspace(lib.bpx_conv3d_bwd_fused_workspace(B, D, H, W, x.C, Cout), self._dev)
                    L.check(lib.bpx_conv3d_bwd_fused(self.bdt, B, D, H, W, dy, wt.data_ptr(), x.view(), nrm.rec.data_ptr(), nrm.act, L.tview(g), red.data_ptr(),
                                                     G[wk].data_ptr(), G[bk].data_ptr(), None, ws.data_ptr(), ws.numel(), self._st))
                elif nrm is not None:
                    rt = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, x.C)
                    red = torch.empty((B, rt, 2, x.C), dtype=torch.float32, device=self._dev)
                    L.check(lib.bpx_conv3d_dgrad(self.bdt, B, D, H, W, dy, wt.data_ptr(), x.view(), nrm.rec.data_ptr(), nrm.act, L.tview(g), red.data_ptr(), self._st))
                if nrm is not None:
                    coef = torch.empty((B, x.C, 4), dtype=torch.float32, device=self._dev)
                    L.check(lib.bpx_norm_bwd_finalize(red.data_ptr(), B, rt, x.C, x.vox, nrm.rec.data_ptr(), nrm.gamma.data_ptr(), L.ptr(nrm.dgamma),
                                                      L.ptr(nrm.dbeta), x.C, coef.data_ptr(), self._st))
                    add, dst, fresh = self._acc_target(x)             # the InstanceNorm-backward affine adds to the gradient x already has
                    L.check(lib.bpx_norm_bwd_apply(self.bdt, B, x.vox, L.tview(g), x.view(), coef.data_ptr(), add, L.tview(dst), self._st))
                    if fresh:
                        x.grad = dst
                    self._keep += [y.grad, g]
                    return
                L.check(lib.bpx_conv3d_dgrad(self.gdt, B, D, H, W, dy, wt.data_ptr(), L.NULL_T, None, 0, L.tview(g), None, self._st))
                self._keep.append(y.grad)
                self._accum(x, g)
            self._tape.append(bwd)
        return y, part, tiles

    def _conv3_c1(self, img: torch.Tensor, S, wk: str, bk: str, Cout: int):
        P, G, B = self._P, self._G, self._B
        D, H, W = S
        y = self._new(S, Cout)
        tiles = lib.bpx_conv3d_c1_stats_tiles(D, H, W)
        part = _Stats.alloc(B, tiles, Cout, self._dev)
        L.check(lib.bpx_conv3d_c1_fwd(self.dt, B, D, H, W, img.data_ptr(), P[wk].data_ptr(), P[bk].data_ptr(), y.view(), part.data_ptr(), self._st))
        if G is not None:
            def bwd():
                wsc = self._workspace(lib.bpx_conv3d_c1_wgrad_workspace(y.C), self._dev)
                L.check(lib.bpx_conv3d_c1_wgrad(self.gdt, B, D, H, W, img.data_ptr(), L.tview(y.grad), G[wk].data_ptr(), G[bk].data_ptr(),
                                                wsc.data_ptr(), wsc.numel(), self._st))
                self._keep.append(y.grad)
            self._tape.append(bwd)
        return y, part, tiles

    def _add_in(self, main: _V, s: _V, rec_s, gk: str, bk: str) -> _V:
        """out = main + IN(s) (the residual sum with the normalised shortcut), one streaming pass."""
        P, G, B = self._P, self._G, self._B
        out = self._new(main.S, main.C)
        scale, shift = rec_s[:, :, 2].contiguous(), rec_s[:, :, 3].contiguous()
        L.check(lib.bpx_channel_affine(self.dt, B, main.vox, main.view(), s.view(), scale.data_ptr(), shift.data_ptr(), out.view(), self._st))
        if G is not None:
            def bwd():
                self._accum(s, self._in_bwd(s, rec_s, 0, out.grad, P[gk], G[gk], G[bk]))
                self._accum(main, out.grad)
            self._tape.append(bwd)
        return out

    def _res_block(self, x: Optional[_V], img, S, prefix: str, first: bool, Cout: int, rec_x=None) -> _V:
        """blocks.py:1304-1378 with a 3x3x3 shortcut + norm.  rec_x: pre-norm record of x (non-first blocks)."""
        P, G = self._P, self._G
        i = 0 if first else 2
        k1, k2 = f"{prefix}.block.{i}.block", f"{prefix}.block.{i + 1}.block"
        vox = S[0] * S[1] * S[2]
        if first:
            h, part, tiles = self._conv3_c1(img, S, f"{k1}.0.weight", f"{k1}.0.bias", Cout)
            s, spart, stiles = self._conv3_c1(img, S, f"{prefix}.shortcut.0.weight", f"{prefix}.shortcut.0.bias", Cout)
        else:
            g0, b0 = f"{prefix}.block.0.weight", f"{prefix}.block.0.bias"
            nrm = _Nrm(rec_x, self.act, P[g0], G[g0] if G is not None else None, G[b0] if G is not None else None)
            h, part, tiles = self._conv3(x, f"{k1}.0.weight", f"{k1}.0.bias", Cout, nrm)
            s, spart, stiles = self._conv3(x, f"{prefix}.shortcut.0.weight", f"{prefix}.shortcut.0.bias", Cout)
        rec_h = self._finalize(part, tiles, Cout, vox, P[f"{k1}.1.weight"], P[f"{k1}.1.bias"])
        nrm_h = _Nrm(rec_h, self.act, P[f"{k1}.1.weight"], G[f"{k1}.1.weight"] if G is not None else None, G[f"{k1}.1.bias"] if G is not None else None)
        main, _, _ = self._conv3(h, f"{k2}.0.weight", f"{k2}.0.bias", Cout, nrm_h, want_stats=False)
        rec_s = self._finalize(spart, stiles, Cout, vox, P[f"{prefix}.shortcut.1.weight"], P[f"{prefix}.shortcut.1.bias"])
        return self._add_in(main, s, rec_s, f"{prefix}.shortcut.1.weight", f"{prefix}.shortcut.1.bias")

    def _sqex(self, x: _V, prefix: str) -> _V:
        """blocks.py:1119-1191: x * sigmoid(W2 relu(W1 mean(x))) - the gate on the pooled vector is one small kernel each way."""
        P, G, B = self._P, self._G, self._B
        C = x.C
        k1, k2 = f"{prefix}.excitation.0.weight", f"{prefix}.excitation.2.weight"
        w1, w2 = P[k1], P[k2]
        R = w1.shape[0]
        part, tiles = self._tensor_part(x)
        s = torch.empty((B, C), dtype=torch.float32, device=self._dev)
        sv = torch.empty((B, C + 2 * R), dtype=torch.float32, device=self._dev)
        L.check(lib.bpx_gate_mlp_fwd(part.data_ptr(), B, tiles, C, x.vox, w1.data_ptr(), None, w2.data_ptr(), None, R, self.relu, s.data_ptr(), sv.data_ptr(),
                                     self._st))
        out = self._new(x.S, C)
        L.check(lib.bpx_channel_affine(self.dt, B, x.vox, L.NULL_T, x.view(), s.data_ptr(), None, out.view(), self._st))
        if G is not None:
            def bwd():
                nt = lib.bpx_norm_act_tiles(self.gdt, x.vox, C)
                dpart = torch.empty((B, nt, C), dtype=torch.float32, device=self._dev)
                L.check(lib.bpx_dot_stats(self.bdt, B, x.vox, L.tview(out.grad), x.view(), dpart.data_ptr(), self._st))
                off = torch.empty((B, C), dtype=torch.float32, device=self._dev)         # d mean / voxels -> every voxel of the channel
                L.check(lib.bpx_gate_mlp_bwd(dpart.data_ptr(), B, nt, C, x.vox, s.data_ptr(), sv.data_ptr(), w1.data_ptr(), w2.data_ptr(), R, self.relu,
                                             G[k1].data_ptr(), None, G[k2].data_ptr(), None, off.data_ptr(), self._st))
                add, dst, fresh = self._acc_target(x)
                L.check(lib.bpx_channel_affine(self.gdt, B, x.vox, add, L.tview(out.grad), s.data_ptr(), off.data_ptr(), L.tview(dst), self._st))
                if fresh:
                    x.grad = dst
                self._keep += [out.grad, off, dpart]
            self._tape.append(bwd)
        return out

    def _maxpool(self, x: _V, sz: int) -> _V:
        G, B = self._G, self._B
        D, H, W = x.S
        out = self._new((D // sz, H // 2, W // 2), x.C)
        ppart = _Stats.alloc(B, lib.bpx_maxpool3d_stats_tiles(self.dt, D, H, W, sz, x.C), x.C, self._dev)      # statistics unused here
        L.check(lib.bpx_maxpool3d_fwd(self.dt, B, D, H, W, sz, x.view(), out.view(), ppart.data_ptr(), self._st))
        if G is not None:
            def bwd():
                if x.grad is None:
                    x.grad = torch.zeros((B,) + x.S + (x.C,), dtype=self.gdtype, device=self._dev)
                L.check(lib.bpx_maxpool3d_bwd(self.bdt, B, D, H, W, sz, x.view(), L.tview(out.grad), L.tview(x.grad), L.tview(x.grad), self._st))
                self._keep.append(out.grad)
            self._tape.append(bwd)
        return out

    def _conv1x1(self, x: _V, w: torch.Tensor, b: torch.Tensor, on_grads: Callable[[torch.Tensor, torch.Tensor], None]) -> _V:
        """1x1x1 convolution (Cin, Cout multiples of 16) as a GEMM over voxels; ``on_grads(dW, db)`` receives the parameter gradients."""
        G, B = self._G, self._B
        Cout, Cin = w.shape[0], w.shape[1]
        y = self._new(x.S, Cout)
        wp = self._pack(w, L.PK_DENSE, Cin, Cout, False)
        L.check(lib.bpx_conv1x1_fwd(self.dt, B, x.vox, x.view(), wp.data_ptr(), b.data_ptr(), L.NULL_T, L.NULL_T, None, L.NULL_T, y.view(), self._st))
        if G is not None:
            def bwd():
                dw = torch.zeros((Cout, Cin, 1, 1, 1), dtype=torch.float32, device=self._dev)
                db = torch.zeros((Cout,), dtype=torch.float32, device=self._dev)
                self._wgrad(B, x.S, x.view(), None, 0, L.tview(y.grad), 1, dw, db, self._st, self._dev)
                wt = self._pack(w, L.PK_DENSE_T, Cin, Cout, False)
                add, dst, fresh = self._acc_target(x)
                L.check(lib.bpx_conv1x1_fwd(self.gdt, B, x.vox, L.tview(y.grad), wt.data_ptr(), None, L.NULL_T, L.NULL_T, None, add, L.tview(dst), self._st))
                if fresh:
                    x.grad = dst
                self._keep.append(y.grad)
                self._late.append(lambda: on_grads(dw, db))              # after the deferred wgrad reductions have run
            self._tape.append(bwd)
        return y

    def _aspp(self, x: _V, prefix: str, Cout: int) -> _V:
        """heads.py:13-133: concat_j IN(ReLU(conv_{d_j}(x))) -> 1x1x1 conv."""
        P, G, B = self._P, self._G, self._B
        rates = list(self.pp.rates)
        cat = self._new(x.S, Cout * len(rates))
        branches = []
        for j, d in enumerate(rates):
            wk, bk = f"{prefix}.aspp_block{j + 1}.0.weight", f"{prefix}.aspp_block{j + 1}.0.bias"
            gk, bek = f"{prefix}.aspp_block{j + 1}.2.weight", f"{prefix}.aspp_block{j + 1}.2.bias"
            if all(d >= n for n in x.S):
                # every off-centre tap of a rate-d kernel lands in the zero padding of a volume that is at most d wide (the bridge of
                # cfg 4: 5^3 at rates 6/12/18): the convolution IS the 1x1x1 convolution with the centre tap, and the other 26 taps
                # have an exactly-zero gradient, as in the reference
                wc = P[wk][:, :, 1:2, 1:2, 1:2].contiguous()
                raw = self._new(x.S, Cout)
                L.check(lib.bpx_conv1x1_fwd(self.dt, B, x.vox, x.view(), self._pack(wc, L.PK_DENSE, x.C, Cout, False).data_ptr(), P[bk].data_ptr(),
                                            L.NULL_T, L.NULL_T, None, L.NULL_T, raw.view(), self._st))
                r = self._new(x.S, Cout)
                L.check(lib.bpx_norm_act_fwd(self.dt, B, x.vox, raw.view(), self._ident_rec(Cout).data_ptr(), self.relu, r.view(), self._st))
                rec = self._stats_rec(r, P[gk], P[bek])
                L.check(lib.bpx_norm_act_fwd(self.dt, B, x.vox, r.view(), rec.data_ptr(), 0, L.tview(cat.buf, j * Cout, Cout), self._st))
                branches.append((0, wc, None, raw, r, rec, wk, bk, gk, bek))
                continue
            # the d^3 sub-lattices of a sample side by side in one volume with zero separator planes (dilation.packed_tables): one ordinary
            # convolution per branch on full tiles instead of B * d^3 convolutions of n^3 <= 14^3 voxels
            tables = self._c(("packed", x.S, d), lambda: torch.from_numpy(dilation.packed_tables(x.S, d)).to(self._dev))
            xs = dilation.space_to_packed(x.buf, d, tables)                                  # (B, pz, py, px, Cin)
            xsv = _V(xs, tuple(xs.shape[1:4]))
            self._G, keepG = None, self._G                                                   # the packed conv is differentiated by hand below
            ys, _, _ = self._conv3(xsv, wk, bk, Cout, None, want_stats=False, batch=B)
            self._G = keepG
            raw = _V(dilation.packed_to_space(ys.buf, d, x.S, tables), x.S)
            r = self._new(x.S, Cout)
            L.check(lib.bpx_norm_act_fwd(self.dt, B, x.vox, raw.view(), self._ident_rec(Cout).data_ptr(), self.relu, r.view(), self._st))
            rec = self._stats_rec(r, P[gk], P[bek])
            L.check(lib.bpx_norm_act_fwd(self.dt, B, x.vox, r.view(), rec.data_ptr(), 0, L.tview(cat.buf, j * Cout, Cout), self._st))
            branches.append((d, tables, xsv, raw, r, rec, wk, bk, gk, bek))
        if G is not None:
            def bwd():
                for j, (d, tables, xsv, raw, r, rec, wk, bk, gk, bek) in enumerate(branches):
                    dslice = torch.empty((B,) + x.S + (Cout,), dtype=self.gdtype, device=self._dev)   # dense copy of the branch's slice of d(cat)
                    L.check(lib.bpx_norm_act_fwd(self.gdt, B, x.vox, L.tview(cat.grad, j * Cout, Cout), self._ident_rec(Cout).data_ptr(), 0, L.tview(dslice), self._st))
                    dr = self._in_bwd(r, rec, 0, dslice, P[gk], G[gk], G[bek])
                    draw = self._act_bwd(raw, self.relu, dr)
                    if d == 0:                                                               # centre-tap branch (tables = the centre tap)
                        dwc = torch.zeros((Cout, x.C, 1, 1, 1), dtype=torch.float32, device=self._dev)
                        dbc = torch.zeros((Cout,), dtype=torch.float32, device=self._dev)
                        self._wgrad(B, x.S, x.view(), None, 0, L.tview(draw), 1, dwc, dbc, self._st, self._dev)
                        add, dst, fresh = self._acc_target(x)
                        L.check(lib.bpx_conv1x1_fwd(self.gdt, B, x.vox, L.tview(draw), self._pack(tables, L.PK_DENSE_T, x.C, Cout, False).data_ptr(), None,
                                                    L.NULL_T, L.NULL_T, None, add, L.tview(dst), self._st))
                        if fresh:
                            x.grad = dst

                        def late(wk=wk, bk=bk, dwc=dwc, dbc=dbc):                            # after the deferred wgrad reductions have run
                            G[wk].zero_()
                            G[wk][:, :, 1:2, 1:2, 1:2] = dwc
                            G[bk].copy_(dbc)
                        self._late.append(late)
                        self._keep += [draw, dslice]
                        continue
                    dys = dilation.space_to_packed(draw, d, tables)                          # zero on the separators: they add nothing to dW, db
                    nb = dys.shape[0]
                    self._wgrad(nb, xsv.S, xsv.view(), None, 0, L.tview(dys), 3, G[wk], G[bk], self._st, self._dev)
                    wt = self._pack(P[wk], L.PK_K3_T, x.C, Cout, False)
                    gs = torch.empty(tuple(xsv.buf.shape), dtype=self.gdtype, device=self._dev)
                    L.check(lib.bpx_conv3d_dgrad(self.gdt, nb, xsv.S[0], xsv.S[1], xsv.S[2], L.tview(dys), wt.data_ptr(), L.NULL_T, None, 0, L.tview(gs), None, self._st))
                    self._keep += [draw, dys, gs]
                    self._accum(x, dilation.packed_to_space(gs, d, x.S, tables))
                self._keep.append(cat.grad)
            self._tape.append(bwd)                                                           # runs AFTER the 1x1 conv's closure (pushed next)
        ow, ob = f"{prefix}.output.weight", f"{prefix}.output.bias"

        def on_grads(dw, db):
            G[ow].copy_(dw)
            G[ob].copy_(db)
        return self._conv1x1(cat, P[ow], P[ob], on_grads)

    def _attention(self, x1: _V, x2: _V, prefix: str, pool_sz: int) -> _V:
        """blocks.py:2168-2298: gate = conv1(ReLU(IN(maxpool(conv3(ReLU(IN(x1)))) + conv3(ReLU(IN(x2)))))), out = gate * x2."""
        P, G, B = self._P, self._G, self._B
        C = x2.C

        def nrm_of(v, name):
            gk, bk = f"{prefix}.{name}.0.weight", f"{prefix}.{name}.0.bias"
            return _Nrm(self._stats_rec(v, P[gk], P[bk]), self.relu, P[gk], G[gk] if G is not None else None, G[bk] if G is not None else None)

        e_full, _, _ = self._conv3(x1, f"{prefix}.conv_encoder.2.weight", f"{prefix}.conv_encoder.2.bias", C, nrm_of(x1, "conv_encoder"), want_stats=False)
        e = self._maxpool(e_full, pool_sz)
        d, _, _ = self._conv3(x2, f"{prefix}.conv_decoder.2.weight", f"{prefix}.conv_decoder.2.bias", C, nrm_of(x2, "conv_decoder"), want_stats=False)
        sm = self._new(x2.S, C)
        L.check(lib.bpx_channel_affine(self.dt, B, x2.vox, e.view(), d.view(), self._ones_bc(C).data_ptr(), None, sm.view(), self._st))
        gk, bk = f"{prefix}.conv_attn.0.weight", f"{prefix}.conv_attn.0.bias"
        rec = self._stats_rec(sm, P[gk], P[bk])
        m = self._new(x2.S, C)
        L.check(lib.bpx_norm_act_fwd(self.dt, B, x2.vox, sm.view(), rec.data_ptr(), self.relu, m.view(), self._st))
        if G is not None:
            def bwd_sum():
                dsm = self._in_bwd(sm, rec, self.relu, m.grad, P[gk], G[gk], G[bk])
                self._accum(e, dsm)
                self._accum(d, dsm)                                   # shared, read-only from here on
                e.grad_shared = d.grad_shared = e.grad is d.grad
            self._tape.append(bwd_sum)
        wk, wbk = f"{prefix}.conv_attn.2.weight", f"{prefix}.conv_attn.2.bias"
        w16 = torch.zeros((16, C, 1, 1, 1), dtype=torch.float32, device=self._dev)
        b16 = torch.zeros((16,), dtype=torch.float32, device=self._dev)
        w16[0] = P[wk][0]
        b16[0] = P[wbk][0]

        def on_grads(dw, db):
            G[wk].copy_(dw[:1])
            G[wbk].copy_(db[:1])
        a16 = self._conv1x1(m, w16, b16, on_grads)
        out = self._new(x2.S, C)
        L.check(lib.bpx_gate_mul_fwd(self.dt, B * x2.vox, a16.view(), x2.view(), out.view(), self._st))
        if G is not None:
            def bwd_gate():
                dx2 = torch.empty((B,) + x2.S + (C,), dtype=self.gdtype, device=self._dev)
                a16.grad = torch.empty((B,) + x2.S + (16,), dtype=self.gdtype, device=self._dev)
                L.check(lib.bpx_gate_mul_bwd(self.bdt, B * x2.vox, L.tview(out.grad), a16.view(), x2.view(), L.tview(dx2), a16.grad.data_ptr(), self._st))
                self._keep.append(out.grad)
                self._accum(x2, dx2)
            self._tape.append(bwd_gate)
        return out

    def _up_block(self, x: _V, bridge: _V, j: int, sz: int, Cout: int) -> _V:
        """blocks.py:1603-1655: cat([ConvTranspose(x), bridge]) -> residual block."""
        P, G, B = self._P, self._G, self._B
        Cup, Cb = x.C, bridge.C
        Dl, Hl, Wl = x.S
        S = bridge.S
        vox = S[0] * S[1] * S[2]
        cat = self._new(S, Cup + Cb)
        wk, bk = f"up_paths.0.{j}.up.weight", f"up_paths.0.{j}.up.bias"
        wp = self._pack(P[wk], L.PK_CT if sz == 2 else L.PK_CT4, Cup, Cup, False)
        utiles = lib.bpx_convT3d_stats_tiles(Dl, Hl, Wl, sz)
        upart = _Stats.alloc(B, utiles, Cup, self._dev)
        L.check(lib.bpx_convT3d_k2s2_fwd(self.dt, B, Dl, Hl, Wl, sz, x.view(), wp.data_ptr(), P[bk].data_ptr(), L.tview(cat.buf, 0, Cup), upart.data_ptr(), self._st))
        L.check(lib.bpx_norm_act_fwd(self.dt, B, vox, bridge.view(), self._ident_rec(Cb).data_ptr(), 0, L.tview(cat.buf, Cup, Cb), self._st))   # the skip copy
        pre = f"up_paths.0.{j}.conv_block"
        g0, be0 = P[f"{pre}.block.0.weight"], P[f"{pre}.block.0.bias"]
        rec = _recs(B, Cup + Cb, self._dev)
        self._finalize(upart, utiles, Cup, vox, g0[:Cup], be0[:Cup], rec, Cup + Cb, 0)
        bpart, btiles = self._tensor_part(bridge)
        self._finalize(bpart, btiles, Cb, vox, g0[Cup:], be0[Cup:], rec, Cup + Cb, Cup)
        if G is not None:
            def bwd():                                                 # runs after the residual block's closures: cat.grad is complete
                dcat = cat.grad
                self._accum(bridge, dcat, Cup)
                ws = self._workspace(lib.bpx_convT3d_k2s2_wgrad_workspace(B, Dl, Hl, Wl, sz, Cup, Cup), self._dev)
                L.check(lib.bpx_convT3d_k2s2_wgrad(self.bdt, B, Dl, Hl, Wl, sz, x.view(), L.tview(dcat, 0, Cup), G[wk].data_ptr(), G[bk].data_ptr(),
                                                   ws.data_ptr(), ws.numel(), self._st))
                wt = self._pack(P[wk], L.PK_CT_T if sz == 2 else L.PK_CT4_T, Cup, Cup, False)
                g = torch.empty((B,) + x.S + (Cup,), dtype=self.gdtype, device=self._dev)
                L.check(lib.bpx_convT3d_k2s2_dgrad(self.gdt, B, Dl, Hl, Wl, sz, L.tview(dcat, 0, Cup), wt.data_ptr(), L.tview(g), self._st))
                self._keep.append(dcat)
                self._accum(x, g)
            self._tape.append(bwd)
        return self._res_block(cat, None, S, pre, False, Cout, rec_x=rec)

    # ------------------------------------------------------------------------------------------------------------------
    def forward(self, P: Dict[str, torch.Tensor], x: torch.Tensor, head_act: int = 0, save: bool = False, cache_weights: bool = False):
        """x: (B,1,Z,Y,X) fp32.  Returns logits (B, sum(out_ch), Z, Y, X) fp32 planar and the saved context (or None)."""
        cfg = self.pp
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] == 1
        B, _, D0, H0, W0 = x.shape
        fm, depth, zd = list(cfg.feature_maps), cfg.depth, cfg.z_down
        zdiv = 1
        for v in zd[1:]:
            zdiv *= v
        if D0 % zdiv or H0 % (2 ** depth) or W0 % (2 ** depth):
            raise ValueError(f"patch {D0, H0, W0} must be divisible by {(zdiv, 2 ** depth, 2 ** depth)} (DATA.PATCH_SIZE rule, check_configuration.py:3156-3202)")
        self._B, self._dev, self._st, self._P = B, x.device, L.stream_ptr(), P
        self._begin_recorded_packs(P, save, x.device, cache_weights)   # one batched weight-pack launch from the second step on
        self._tape: List[Callable[[], None]] = []
        self._late: List[Callable[[], None]] = []
        self._keep = []
        self._G = None
        if save:
            names = list(P.keys())
            flat = torch.zeros(sum(P[n].numel() for n in names), dtype=torch.float32, device=x.device)
            self._G, o = {}, 0
            for n in names:
                self._G[n] = flat[o:o + P[n].numel()].view(P[n].shape)
                o += P[n].numel()
        img = x.reshape(B, D0, H0, W0).contiguous()
        S = [(D0, H0, W0)]
        for i in range(1, depth + 1):
            S.append((S[i - 1][0] // zd[i], S[i - 1][1] // 2, S[i - 1][2] // 2))
        # ---------------- encoder (resunet++.py:435-444): level 0 is not pooled, the last level has no SE block ----------------
        blocks: List[_V] = []
        cur: Optional[_V] = None
        for i in range(depth + 1):
            if i == 0:
                cur = self._res_block(None, img, S[0], "down_path.0", True, fm[0])
            else:
                pre = f"down_path.{i}"
                rec = self._stats_rec(cur, P[f"{pre}.block.0.weight"], P[f"{pre}.block.0.bias"])
                cur = self._res_block(cur, None, cur.S, pre, False, fm[i], rec_x=rec)
            if i < depth:
                cur = self._sqex(cur, f"sqex_blocks.{i}")
            if i != 0:
                cur = self._maxpool(cur, zd[i])
            blocks.append(cur)
        cur = self._aspp(cur, "aspp_bridge", fm[-1])
        # ---------------- decoder ---------------------------------------------------------------------------------------------
        for j in range(depth):
            i = depth - 1 - j
            cur = self._attention(blocks[-j - 2], cur, f"attentions.0.{j}", zd[i + 1])
            cur = self._up_block(cur, blocks[-j - 2], j, zd[i + 1], fm[i + 1])
        feat = self._aspp(cur, "aspp_out.0", fm[0])
        # ---------------- heads -----------------------------------------------------------------------------------------------
        n_out = sum(cfg.out_channels)
        hw = torch.cat([P[f"heads.{h}.weight"].reshape(-1, fm[0]) for h in range(len(cfg.out_channels))], 0).contiguous()
        hb = torch.cat([P[f"heads.{h}.bias"] for h in range(len(cfg.out_channels))], 0).contiguous()
        vox0 = D0 * H0 * W0
        logits = torch.empty((B, n_out, D0, H0, W0), dtype=torch.float32, device=x.device)
        L.check(lib.bpx_head_fwd(self.dt, vox0, B, feat.view(), hw.data_ptr(), hb.data_ptr(), n_out, head_act, logits.data_ptr(), n_out * vox0, vox0, self._st))
        ctx = None
        if save:
            # the packed operands and the parameter versions they were made from travel with the context: a later forward() replaces the
            # engine's own map, and an optimizer step between this forward and its backward would leave the map stale
            ctx = dict(tape=self._tape, late=self._late, G=self._G, feat=feat, hw=hw, B=B, S0=S[0], P=P, prepacked=self._prepacked,
                       pack_names=self._pack_names, pack_seen=self._pack_seen, versions={n: t._version for n, t in P.items()})
        self._tape, self._late, self._G = [], [], None
        self._keep = []
        return logits, ctx

    def backward(self, P: Dict[str, torch.Tensor], ctx, dlogits: torch.Tensor) -> Dict[str, torch.Tensor]:
        cfg = self.pp
        B, feat, G = ctx["B"], ctx["feat"], ctx["G"]
        self._B, self._dev, self._st, self._P, self._G = B, dlogits.device, L.stream_ptr(), ctx["P"], G
        self._late = ctx["late"]
        self._keep = []
        stale = [n for n, v in ctx["versions"].items() if ctx["P"][n]._version != v]
        if stale:
            raise RuntimeError(f"ResUNet++ backward: {len(stale)} parameters (first: {stale[0]}) were modified in place after the forward pass of this context")
        self._prepacked, self._pack_names, self._pack_seen = ctx["prepacked"], ctx["pack_names"], ctx["pack_seen"]
        fm = list(cfg.feature_maps)
        n_out = sum(cfg.out_channels)
        D0, H0, W0 = ctx["S0"]
        vox0 = D0 * H0 * W0
        self._deferred = True
        L.check(lib.bpx_wgrad_defer_begin())
        try:
            dl = dlogits.contiguous().float()
            feat.grad = torch.empty((B, D0, H0, W0, fm[0]), dtype=self.gdtype, device=self._dev)
            hwg = torch.zeros((n_out, fm[0]), dtype=torch.float32, device=self._dev)
            hbg = torch.zeros((n_out,), dtype=torch.float32, device=self._dev)
            hws = self._workspace(lib.bpx_head_bwd_workspace(fm[0], n_out), self._dev)
            L.check(lib.bpx_head_bwd(self.bdt, vox0, B, feat.view(), ctx["hw"].data_ptr(), n_out, dl.data_ptr(), n_out * vox0, vox0, L.tview(feat.grad),
                                     hwg.data_ptr(), hbg.data_ptr(), hws.data_ptr(), hws.numel(), self._st))
            o = 0
            for h, oc in enumerate(cfg.out_channels):
                G[f"heads.{h}.weight"].copy_(hwg[o:o + oc].view(G[f"heads.{h}.weight"].shape))
                G[f"heads.{h}.bias"].copy_(hbg[o:o + oc])
                o += oc
            for fn in reversed(ctx["tape"]):
                fn()
        finally:
            self._deferred = False
            L.check(lib.bpx_wgrad_defer_flush(self._st))
        for fn in self._late:                                          # gradients of padded / re-shaped weights: copied after the flush wrote them
            fn()
        self._keep, self._late, self._G = [], [], None
        ctx["tape"] = []
        return G
