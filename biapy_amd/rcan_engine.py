"""Executor of the RCAN trunk (3D, no up-scaling layer) on the MI355X kernels (cfg 5 family).

Host side of ``biapy/models/rcan.py`` (``rcan.forward`` :335-347, ``RG``, ``RCAB_rcan``, ``ChannelAttention``) and of its autograd
graph, from the same C-ABI kernels as the U-Nets:

  * every 3x3x3 convolution is ``bpx_conv3d_fwd`` / ``_dgrad`` / ``_wgrad``; the SiLU between the two convolutions of an RCAB is the
    consumer's prologue (an identity normalisation record + activation), so the activated tensor never materialises;
  * the residual additions of ``RG`` (``x + module(x)``) and of the trunk (``x += residual``) are extra K-steps of the producing
    convolution: its fused 1x1x1 shortcut operand with an identity matrix;
  * channel attention: the global average pool comes out of the second convolution's statistics epilogue, the two 1x1 "convs"
    on the pooled (B, C) vector are one small kernel each way (``bpx_gate_mlp_fwd`` / ``_bwd``: pooled mean -> SiLU MLP -> sigmoid and
    its hand-written gradient; they were ~8 + ~15 PyTorch launches per RCAB, which bound the 200-RCAB trunk), the recalibration and the
    RCAB residual are one streaming pass (``bpx_channel_affine``: y = x + s[n,c] * h); the backward needs one reduction
    (``bpx_dot_stats``: ds[n,c] = sum dy*h) and the same affine pass (dh = s*dy + dmean/voxels);
  * the last convolution (filters -> out_channels <= 4) runs with its output channels zero-padded to 16 and the head kernel picks
    the real ones and applies the output activation.

The reference's 3-D up-scaling branch (``nn.PixelShuffle`` on 5-D tensors) raises; ``scale`` > 0 runs the DEFINED 3-D form (rcan.py docstring):
forward = ``bpx_conv3d_fwd_shuffle``, backward (round 4) = the shuffle's adjoint (a gather of the sub-positions into channel blocks) + the conv's
own ``bpx_conv3d_wgrad`` / ``_dgrad`` per block of sub-positions on the low-resolution grid.
Weights are packed with one launch per step from the second step on (``ResUNetEngine._begin_recorded_packs``).
"""
from __future__ import annotations

import os
from typing import Dict

import torch

from . import _lib as L
from .engine import NetConfig, ResUNetEngine, _Stats

lib = L.lib



# ---- x scale stage: orders of the 16 s^3 conv channels (pure tensor algebra, checked on the CPU against autograd through the oracle's pixel_shuffle3d) ----
def rows_to_subposition_major(w: torch.Tensor, b: torch.Tensor, Fc: int, s3: int):
    """PyTorch's pixel-shuffle channel order [channel][sub-position] -> the kernels' [sub-position][channel] (weight rows and bias alike)."""
    return (w.reshape(Fc, s3, *w.shape[1:]).transpose(0, 1).reshape(s3 * Fc, *w.shape[1:]).contiguous(), b.reshape(Fc, s3).t().reshape(-1).contiguous())


def rows_from_subposition_major(w: torch.Tensor, b: torch.Tensor, Fc: int, s3: int):
    """Inverse of :func:`rows_to_subposition_major` (what the stage's weight / bias gradients go through)."""
    return (w.reshape(s3, Fc, *w.shape[1:]).transpose(0, 1).reshape(Fc * s3, *w.shape[1:]), b.reshape(s3, Fc).t().reshape(-1))


def gather_subpositions(d_up: torch.Tensor, s: int) -> torch.Tensor:
    """Adjoint of the 3-D pixel shuffle on channels-last tensors: (B, sD, sH, sW, Fc) -> (B, D, H, W, s^3, Fc), sub-position (a, b, e) of voxel
    (z, y, x) = voxel (s z + a, s y + b, s x + e) of the fine grid, sub-positions in the order (a s + b) s + e (rcan.py docstring)."""
    B, Ds, Hs, Ws, Fc = d_up.shape
    D, H, W = Ds // s, Hs // s, Ws // s
    return d_up.view(B, D, s, H, s, W, s, Fc).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(B, D, H, W, s ** 3, Fc)


class RCANEngine(ResUNetEngine):
    def __init__(self, num_channels: int, filters: int, num_rg: int, num_rcab: int, reduction: int, out_channels: int,
                 dtype: torch.dtype = torch.bfloat16, scale: int = 0):
        if num_channels != 1:
            raise NotImplementedError("RCANEngine: one input channel (the first layer kernel is the Cin = 1 one)")
        if filters not in (16, 32) or not 1 <= out_channels <= 4:
            raise NotImplementedError("RCANEngine: filters must be 16 or 32 and out_channels <= 4 (head kernel)")
        super().__init__(NetConfig(in_ch=1, feature_maps=[filters, filters], out_channels=(out_channels,), activation="silu"), dtype)
        self.Fc, self.num_rg, self.num_rcab, self.red, self.n_out = filters, num_rg, num_rcab, max(1, filters // reduction), out_channels
        self.silu = L.ACT["silu"]
        self.up_group = int(os.environ.get("BPX_RCAN_UP_GROUP", "8"))   # sub-positions per backward call of the x scale stage (8: 128-channel blocks; 1: 16-channel blocks)
        self.scale = int(scale)          # > 0: conv(filters -> filters * scale^3) + 3-D pixel shuffle in front of the last conv (rcan.py:344-345)
        if self.scale and (filters != 16 or self.scale not in (2, 3, 4) or dtype == torch.float32):
            raise NotImplementedError("RCANEngine: the up-scaling stage needs 16 filters, scale 2..4 and 16-bit storage (bpx_conv3d_fwd_shuffle)")

    # ---- small helpers -----------------------------------------------------------------------------------------------------
    def _consts(self, B, dev):
        key = (B, str(dev))
        if getattr(self, "_ckey", None) != key:
            Fc = self.Fc
            rec = torch.zeros((B, Fc, 4), dtype=torch.float32, device=dev)
            rec[..., 1] = 1.0
            rec[..., 2] = 1.0                                              # identity record: scale 1, shift 0 -> prologue = activation only
            eye = torch.eye(Fc, dtype=torch.float32, device=dev).reshape(Fc, Fc, 1, 1, 1).contiguous()
            self._c = dict(rec=rec, eye_packed=self._pack(eye, L.PK_K1, Fc, Fc, False), zeros=torch.zeros(Fc, dtype=torch.float32, device=dev),
                           ones_bc=torch.ones((B, Fc), dtype=torch.float32, device=dev), ones=torch.ones(Fc, dtype=torch.float32, device=dev))
            self._ckey = key
        return self._c

    def _conv(self, B, S, x, rec, act, w, b, y, part=None, sc=None):
        D, H, W = S
        c = self._c
        wp = self._pack(w, L.PK_K3, self.Fc, w.shape[0], False)
        if sc is None:
            L.check(lib.bpx_conv3d_fwd(self.dt, B, D, H, W, L.tview(x), L.ptr(rec), act, wp.data_ptr(), b.data_ptr(), L.NULL_T, None, None,
                                       L.tview(y), L.ptr(part), L.stream_ptr()))
        else:                                                                # + identity shortcut: y = conv(x) + sc
            L.check(lib.bpx_conv3d_fwd(self.dt, B, D, H, W, L.tview(x), L.ptr(rec), act, wp.data_ptr(), b.data_ptr(), L.tview(sc),
                                       c["eye_packed"].data_ptr(), c["zeros"].data_ptr(), L.tview(y), L.ptr(part), L.stream_ptr()))

    def _gate_names(self, p):
        return [f"{p}.module.3.module.1.weight", f"{p}.module.3.module.1.bias", f"{p}.module.3.module.3.weight", f"{p}.module.3.module.3.bias"]

    # ---- forward -----------------------------------------------------------------------------------------------------------
    def forward(self, P: Dict[str, torch.Tensor], x: torch.Tensor, head_act: int = 0, save: bool = False, cache_weights: bool = False):
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] == 1
        B, _, D, H, W = x.shape
        S, vox, Fc, T, dev, st = (D, H, W), D * H * W, self.Fc, self.dtype, x.device, L.stream_ptr()
        self._begin_recorded_packs(P, save, dev, cache_weights)
        c = self._consts(B, dev)
        img = x.reshape(B, D, H, W).contiguous()

        def buf(C=Fc):
            return torch.empty((B, D, H, W, C), dtype=T, device=dev)

        f0 = buf()
        part = _Stats.alloc(B, lib.bpx_conv3d_c1_stats_tiles(D, H, W), Fc, dev)
        L.check(lib.bpx_conv3d_c1_fwd(self.dt, B, D, H, W, img.data_ptr(), P["sf.weight"].data_ptr(), P["sf.bias"].data_ptr(), L.tview(f0), part.data_ptr(), st))
        tiles = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, Fc)
        cur, groups = f0, []
        for g in range(self.num_rg):
            xg, z, blocks = cur, cur, []
            for r in range(self.num_rcab):
                p = f"rgs.{g}.module.{r}"
                h1, h2 = buf(), buf()
                self._conv(B, S, z, None, 0, P[f"{p}.module.0.weight"], P[f"{p}.module.0.bias"], h1)
                part2 = _Stats.alloc(B, tiles, Fc, dev)
                self._conv(B, S, h1, c["rec"], self.silu, P[f"{p}.module.2.weight"], P[f"{p}.module.2.bias"], h2, part=part2)
                names = self._gate_names(p)                                   # s = sigmoid(W2 SiLU(W1 mean + b1) + b2) from the statistics partials
                sd = torch.empty((B, Fc), dtype=torch.float32, device=dev)
                sv = torch.empty((B, Fc + 2 * self.red), dtype=torch.float32, device=dev)
                L.check(lib.bpx_gate_mlp_fwd(part2.data_ptr(), B, tiles, Fc, vox, P[names[0]].data_ptr(), P[names[1]].data_ptr(), P[names[2]].data_ptr(),
                                             P[names[3]].data_ptr(), self.red, self.silu, sd.data_ptr(), sv.data_ptr(), st))
                zn = buf()
                L.check(lib.bpx_channel_affine(self.dt, B, vox, L.tview(z), L.tview(h2), sd.data_ptr(), None, L.tview(zn), st))
                blocks.append(dict(p=p, z=z, h1=h1, h2=h2, sd=sd, sv=sv, names=names))
                z = zn
            out = buf()
            pt = f"rgs.{g}.module.{self.num_rcab}"
            self._conv(B, S, z, None, 0, P[f"{pt}.weight"], P[f"{pt}.bias"], out, sc=xg)
            groups.append(dict(pt=pt, z=z, blocks=blocks))
            cur = out
        t = buf()
        self._conv(B, S, cur, None, 0, P["conv1.weight"], P["conv1.bias"], t, sc=f0)
        if self.scale:
            # x scale: the conv's 16 s^3 output channels in the order [sub-position][channel] (PyTorch's pixel-shuffle order is
            # [channel][sub-position]: a permutation of the weight's rows), each 16-channel block stored to its sub-position of the
            # (B, sD, sH, sW, 16) tensor - the 1024-channel tensor of cfg 5 (x4) is never written
            s_ = self.scale
            s3 = s_ ** 3
            wu, bu = rows_to_subposition_major(P["upscale.0.weight"], P["upscale.0.bias"], Fc, s3)
            wpu = self._pack(wu, L.PK_K3, Fc, s3 * Fc, False)
            D, H, W = D * s_, H * s_, W * s_
            up = torch.empty((B, D, H, W, Fc), dtype=T, device=dev)
            L.check(lib.bpx_conv3d_fwd_shuffle(self.dt, B, S[0], S[1], S[2], L.tview(t), None, 0, wpu.data_ptr(), bu.data_ptr(), s_, L.tview(up), st))
            low = dict(S=S, t=t, wu=wu)                                      # what the stage's backward reads: its input and the re-ordered weight
            t, S, vox = up, (D, H, W), D * H * W

            def buf(C=Fc):                                                   # buffers of the high-resolution grid from here on
                return torch.empty((B, D, H, W, C), dtype=T, device=dev)
        # last conv: out_channels (<= 4) zero-padded to 16 output channels; the head kernel picks them and applies the activation
        w2p = torch.zeros((16, Fc, 3, 3, 3), dtype=torch.float32, device=dev)
        b2p = torch.zeros(16, dtype=torch.float32, device=dev)
        w2p[: self.n_out] = P["conv2.weight"]
        b2p[: self.n_out] = P["conv2.bias"]
        o16 = buf(16)
        self._conv(B, S, t, None, 0, w2p, b2p, o16)
        hw = torch.eye(self.n_out, 16, dtype=torch.float32, device=dev).contiguous()
        hb = torch.zeros(self.n_out, dtype=torch.float32, device=dev)
        y = torch.empty((B, self.n_out, D, H, W), dtype=torch.float32, device=dev)
        L.check(lib.bpx_head_fwd(self.dt, vox, B, L.tview(o16), hw.data_ptr(), hb.data_ptr(), self.n_out, head_act, y.data_ptr(), self.n_out * vox, vox, st))
        ctx = dict(B=B, S=S, img=img, f0=f0, groups=groups, last=cur, t=t, o16=o16, w2p=w2p, hw=hw, head_act=head_act, low=low if self.scale else None) if save else None
        return y, ctx

    # ---- backward ------------------------------------------------------------------------------------------------------------
    def backward(self, P: Dict[str, torch.Tensor], ctx, dy_out: torch.Tensor) -> Dict[str, torch.Tensor]:
        """dy_out: gradient of the (linear) output; the head activation must be linear when training through this engine."""
        assert ctx["head_act"] == 0, "train on the linear output (the reference applies its output activation inside the model; pass head_activations=['linear'])"
        B, S = ctx["B"], ctx["S"]
        D, H, W = S
        # T: storage type of the gradient tensors (bf16 in the mixed mode, compute_dtype float16: the forward tensors are fp16 - engine.ResUNetEngine's
        # codes: gdt = kernels on gradient tensors only, bdt = backward kernels that also read a forward tensor)
        vox, Fc, T, dev, st, c = D * H * W, self.Fc, self.gdtype, dy_out.device, L.stream_ptr(), self._c
        self._keep = []
        flat = torch.zeros(sum(p.numel() for p in P.values()), dtype=torch.float32, device=dev)   # ONE fill for the ~1,650 parameter gradients
        G, o = {}, 0
        for n, p in P.items():
            G[n] = flat[o:o + p.numel()].view(p.shape)
            o += p.numel()
        self._deferred = True
        L.check(lib.bpx_wgrad_defer_begin())
        try:
            def buf(C=Fc):
                t_ = torch.empty((B, D, H, W, C), dtype=T, device=dev)
                self._keep.append(t_)
                return t_

            def wgrad(x, rec, act, dy, dw, db):
                self._wgrad(B, S, L.tview(x), rec, act, L.tview(dy), 3, dw, db, st, dev)

            def dgrad(dy, w, t_pre=None, rec=None, act=0):
                g = buf(w.shape[1])
                wt = self._pack(w, L.PK_K3_T, w.shape[1], w.shape[0], False)
                L.check(lib.bpx_conv3d_dgrad(self.bdt if t_pre is not None else self.gdt, B, D, H, W, L.tview(dy), wt.data_ptr(), L.tview(t_pre) if t_pre is not None else L.NULL_T,
                                             L.ptr(rec), act, L.tview(g), None, st))
                return g

            def add(a, b):                                                     # a + b, elementwise
                o = buf()
                L.check(lib.bpx_channel_affine(self.gdt, B, vox, L.tview(a), L.tview(b), c["ones_bc"].data_ptr(), None, L.tview(o), st))
                return o

            # head (linear): gradient of the padded 16-channel tensor
            do16 = buf(16)
            hwg = torch.zeros((self.n_out, 16), dtype=torch.float32, device=dev)
            hbg = torch.zeros((self.n_out,), dtype=torch.float32, device=dev)
            dl = dy_out.contiguous().float()
            hws = self._workspace(lib.bpx_head_bwd_workspace(16, self.n_out), dev)
            L.check(lib.bpx_head_bwd(self.bdt, vox, B, L.tview(ctx["o16"]), ctx["hw"].data_ptr(), self.n_out, dl.data_ptr(), self.n_out * vox, vox,
                                     L.tview(do16), hwg.data_ptr(), hbg.data_ptr(), hws.data_ptr(), hws.numel(), st))
            dw16 = torch.zeros((16, Fc, 3, 3, 3), dtype=torch.float32, device=dev)
            db16 = torch.zeros(16, dtype=torch.float32, device=dev)
            wgrad(ctx["t"], None, 0, do16, dw16, db16)
            dt = dgrad(do16, ctx["w2p"])
            if self.scale:
                # x scale stage: dt is the gradient of the shuffled tensor (B, sD, sH, sW, Fc).  Gathering its s^3 sub-positions back into channel
                # blocks ([sub-position][channel], the order the forward kernel takes the weight rows in) IS the pixel shuffle's adjoint; the conv's two
                # gradients then run per group of `up_group` sub-positions on the low-resolution grid: dW / db rows of the group (independent), and
                # the input gradient as the sum of the groups' dgrads (the convolution is linear in its output-channel blocks)
                s_, low = self.scale, ctx["low"]
                s3 = s_ ** 3
                S = low["S"]
                D, H, W = S                                                    # the helpers above work on the low-resolution grid from here on
                vox = D * H * W
                dsub = gather_subpositions(dt, s_)
                wu = low["wu"]
                dwu = torch.zeros_like(wu)
                dbu = torch.zeros(s3 * Fc, dtype=torch.float32, device=dev)
                parts, gs = [], max(1, int(self.up_group))
                for i in range(0, s3, gs):
                    n = min(gs, s3 - i)
                    dyb = dsub[..., i:i + n, :].reshape(B, D, H, W, n * Fc).contiguous()
                    self._keep.append(dyb)
                    wgrad(low["t"], None, 0, dyb, dwu[i * Fc:(i + n) * Fc], dbu[i * Fc:(i + n) * Fc])
                    parts.append(dgrad(dyb, wu[i * Fc:(i + n) * Fc]))
                while len(parts) > 1:                                          # pairwise: log2(groups) roundings of the 16-bit sums
                    parts = [add(parts[j], parts[j + 1]) if j + 1 < len(parts) else parts[j] for j in range(0, len(parts), 2)]
                dt = parts[0]
            # conv1 (+ identity from f0)
            wgrad(ctx["last"], None, 0, dt, G["conv1.weight"], G["conv1.bias"])
            dcur = dgrad(dt, P["conv1.weight"])
            for g in range(self.num_rg - 1, -1, -1):
                grp = ctx["groups"][g]
                pt = grp["pt"]
                wgrad(grp["z"], None, 0, dcur, G[f"{pt}.weight"], G[f"{pt}.bias"])
                dz = dgrad(dcur, P[f"{pt}.weight"])
                for blk in reversed(grp["blocks"]):
                    p = blk["p"]
                    nt = lib.bpx_norm_act_tiles(self.gdt, vox, Fc)
                    dpart = torch.empty((B, nt, Fc), dtype=torch.float32, device=dev)
                    L.check(lib.bpx_dot_stats(self.bdt, B, vox, L.tview(dz), L.tview(blk["h2"]), dpart.data_ptr(), st))
                    nm = blk["names"]
                    off = torch.empty((B, Fc), dtype=torch.float32, device=dev)   # d mean / voxels -> every voxel of the channel
                    L.check(lib.bpx_gate_mlp_bwd(dpart.data_ptr(), B, nt, Fc, vox, blk["sd"].data_ptr(), blk["sv"].data_ptr(), P[nm[0]].data_ptr(),
                                                 P[nm[2]].data_ptr(), self.red, self.silu, G[nm[0]].data_ptr(), G[nm[1]].data_ptr(), G[nm[2]].data_ptr(),
                                                 G[nm[3]].data_ptr(), off.data_ptr(), st))
                    dh2 = buf()
                    L.check(lib.bpx_channel_affine(self.gdt, B, vox, L.NULL_T, L.tview(dz), blk["sd"].data_ptr(), off.data_ptr(), L.tview(dh2), st))
                    self._keep += [off, dpart]
                    wgrad(blk["h1"], c["rec"], self.silu, dh2, G[f"{p}.module.2.weight"], G[f"{p}.module.2.bias"])
                    dh1 = dgrad(dh2, P[f"{p}.module.2.weight"], blk["h1"], c["rec"], self.silu)
                    wgrad(blk["z"], None, 0, dh1, G[f"{p}.module.0.weight"], G[f"{p}.module.0.bias"])
                    dz = add(dz, dgrad(dh1, P[f"{p}.module.0.weight"]))
                dcur = add(dz, dcur)                                           # through the RCABs + the group's identity path
            df0 = add(dcur, dt)                                                # + the trunk's `x += residual`
            wsc = self._workspace(lib.bpx_conv3d_c1_wgrad_workspace(self.Fc), dev)
            L.check(lib.bpx_conv3d_c1_wgrad(self.gdt, B, D, H, W, ctx["img"].data_ptr(), L.tview(df0), G["sf.weight"].data_ptr(), G["sf.bias"].data_ptr(),
                                            wsc.data_ptr(), wsc.numel(), st))
        finally:
            self._deferred = False
            L.check(lib.bpx_wgrad_defer_flush(st))
        G["conv2.weight"].copy_(dw16[: self.n_out])
        G["conv2.bias"].copy_(db16[: self.n_out])
        if self.scale:                                                       # rows back from [sub-position][channel] to PyTorch's [channel][sub-position]
            s3 = self.scale ** 3
            gw, gb = rows_from_subposition_major(dwu, dbu, Fc, s3)
            G["upscale.0.weight"].copy_(gw)
            G["upscale.0.bias"].copy_(gb)
        self._keep = []
        return G
