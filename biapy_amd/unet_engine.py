"""Executor of the plain (non-residual) U-Net, 2D and 3D, on the MI355X kernels (SURVEY.md row U / cfg 1 family).

Host side of biapy/models/unet.py:382-420 and of its autograd graph, built from the same C-ABI kernels as the residual
network (engine.py) plus the materialised InstanceNorm+activation pair (``bpx_norm_act_fwd`` / ``bpx_norm_act_bwd``):

  * inside a ConvBlock (blocks.py:120-167, ``nconvs`` x [Conv -> IN -> act]) the normalisation + activation between two
    convolutions never materialises: the consumer conv applies the producer's (mean, rstd, scale, shift) records while
    staging its halo, exactly as in the residual network;
  * the block OUTPUT ``act(IN(conv))`` feeds MaxPool / ConvTranspose / the head / the skip connection, none of which has a
    fused prologue, so it is written once by ``bpx_norm_act_fwd`` - the skip copy straight into the channel slice
    ``[Cup, Cup+Cskip)`` of the level's concat buffer (``torch.cat([up, bridge], 1)``, blocks.py:666, is a layout decision);
  * ``UpBlock.up`` = ConvTranspose(in -> out) -> IN -> act (blocks.py:602-614): raw transposed-conv output + statistics, then
    ``bpx_norm_act_fwd`` into channels ``[0, Cup)`` of the concat buffer.

2D networks run as 3D tensors with one z-slice: (B,C,Y,X) <-> NDHWC with D = 1, 3x3 kernels zero-padded to 3x3x3
(only the centre z-tap is non-zero), pooling / transposed convolutions with the (1,2,2) variants.  That wastes 2/3 of the
MFMA work of those layers; cfg 1 is the reference's CPU plumbing configuration, a parity case and not a bench line.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .engine import NetConfig, ResUNetEngine, _recs, _Stats, lift_params, unlift_grads

lib = L.lib


@dataclass
class _CB:
    """What the backward needs about one ConvBlock (two convolutions)."""
    prefix: str
    S: Tuple[int, int, int]
    cin: int
    cout: int
    x: Optional[torch.Tensor]          # materialised block input (None: the 1-channel image)
    h: List[torch.Tensor]              # raw conv outputs
    rec: List[torch.Tensor]            # their norm records


class UNetEngine(ResUNetEngine):
    def __init__(self, cfg: NetConfig, ndim: int, dtype: torch.dtype = torch.bfloat16, nconvs: int = 2):
        super().__init__(cfg, dtype)
        assert ndim in (2, 3)
        if nconvs != 2:
            raise NotImplementedError("UNetEngine: conv_layers must be 2 at every level")
        self.ndim = ndim
        self.nconvs = nconvs

    # ---- parameters: 2D weights are lifted to one-z-slice 3D weights (engine.lift_params) -----------------------------------
    def _lift(self, P: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return lift_params(P) if self.ndim == 2 else P

    # ---- forward ------------------------------------------------------------------------------------------------------
    def _conv_block_fwd(self, P, prefix, B, S, cin, cout, x, img, out_view, st, dev) -> _CB:
        D, H, W = S
        vox = D * H * W
        T = self.dtype
        cb = _CB(prefix, S, cin, cout, x, [], [])
        src, src_rec = x, None
        for c in range(self.nconvs):
            kw, kb = f"{prefix}.block.{c}.block.0.weight", f"{prefix}.block.{c}.block.0.bias"
            h = torch.empty((B, D, H, W, cout), dtype=T, device=dev)
            if c == 0 and img is not None:
                tiles = lib.bpx_conv3d_c1_stats_tiles(D, H, W)
                part = _Stats.alloc(B, tiles, cout, dev)
                L.check(lib.bpx_conv3d_c1_fwd(self.dt, B, D, H, W, img.data_ptr(), P[kw].data_ptr(), P[kb].data_ptr(), L.tview(h), part.data_ptr(), st))
            else:
                ci = cin if c == 0 else cout
                tiles = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, cout)
                part = _Stats.alloc(B, tiles, cout, dev)
                wp = self._pack(P[kw], L.PK_K3, ci, cout, False)
                L.check(lib.bpx_conv3d_fwd(self.dt, B, D, H, W, L.tview(src), L.ptr(src_rec), self.act if src_rec is not None else 0,
                                           wp.data_ptr(), P[kb].data_ptr(), L.NULL_T, None, None, L.tview(h), part.data_ptr(), st))
            rec = _recs(B, cout, dev)
            _Stats.finalize(part, B, tiles, cout, vox, P[f"{prefix}.block.{c}.block.1.weight"], P[f"{prefix}.block.{c}.block.1.bias"],
                            rec, cout, 0, st)
            cb.h.append(h)
            cb.rec.append(rec)
            src, src_rec = h, rec
        L.check(lib.bpx_norm_act_fwd(self.dt, B, vox, L.tview(src), src_rec.data_ptr(), self.act, out_view, st))
        return cb

    def forward(self, P: Dict[str, torch.Tensor], x: torch.Tensor, head_act: int = 0, save: bool = False, cache_weights: bool = False):
        """x: (B,C,Y,X) [2D] or (B,C,Z,Y,X) [3D] fp32.  Returns logits in the same channels-first planar layout, and the context."""
        cfg = self.cfg
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == self.ndim + 2
        if self.ndim == 2:
            x = x.unsqueeze(2)
        B, Cin, D0, H0, W0 = x.shape
        assert Cin == cfg.in_ch, f"expected {cfg.in_ch} input channels, got {Cin}"
        Lv = cfg.depth
        zd = cfg.z_down if self.ndim == 3 else (1,) * Lv
        zdiv = 1
        for v in zd:
            zdiv *= v
        if D0 % zdiv or H0 % (2 ** Lv) or W0 % (2 ** Lv):
            raise ValueError(f"patch {D0, H0, W0} must be divisible by {(zdiv, 2 ** Lv, 2 ** Lv)} (DATA.PATCH_SIZE rule, check_configuration.py:3156-3202)")
        dev, st, T = x.device, L.stream_ptr(), self.dtype
        fm = list(cfg.feature_maps)
        self._prepacked = {}
        Pw = self._lift(P)
        if Cin == 1:
            img, cur = x.reshape(B, D0, H0, W0).contiguous(), None
        else:
            img = None
            xin = x.permute(0, 2, 3, 4, 1).contiguous()
            cur = torch.empty(xin.shape, dtype=T, device=dev)
            if T == torch.float32:
                cur.copy_(xin)
            else:
                L.check(lib.bpx_cast(L.F32, xin.data_ptr(), self.dt, cur.data_ptr(), xin.numel(), st))
        S = [(D0, H0, W0)]
        for i in range(Lv):
            S.append((S[i][0] // zd[i], S[i][1] // 2, S[i][2] // 2))

        def buf(i, C):
            return torch.empty((B,) + S[i] + (C,), dtype=T, device=dev)

        cat = [buf(i, 2 * fm[i]) for i in range(Lv)]           # [up (fm[i]) | skip (fm[i])]
        blocks: List[_CB] = []
        # ---------------- encoder ------------------------------------------------------------------
        for i in range(Lv):
            cin = cfg.in_ch if i == 0 else fm[i - 1]
            cb = self._conv_block_fwd(Pw, f"down_path.{i}", B, S[i], cin, fm[i], cur, img if i == 0 else None, L.tview(cat[i], fm[i], fm[i]), st, dev)
            blocks.append(cb)
            D, H, W = S[i]
            pooled = buf(i + 1, fm[i])
            ppart = _Stats.alloc(B, lib.bpx_maxpool3d_stats_tiles(self.dt, D, H, W, zd[i], fm[i]), fm[i], dev)   # statistics unused here
            L.check(lib.bpx_maxpool3d_fwd(self.dt, B, D, H, W, zd[i], L.tview(cat[i], fm[i], fm[i]), L.tview(pooled), ppart.data_ptr(), st))
            cur = pooled
        # ---------------- bottleneck ----------------------------------------------------------------
        bott = buf(Lv, fm[Lv])
        blocks.append(self._conv_block_fwd(Pw, "bottleneck", B, S[Lv], fm[Lv - 1], fm[Lv], cur, None, L.tview(bott), st, dev))
        # ---------------- decoder -------------------------------------------------------------------
        dec_in, ups = bott, []
        for j, i in enumerate(range(Lv - 1, -1, -1)):
            Cl, Cup = fm[i + 1], fm[i]
            Dl, Hl, Wl = S[i + 1]
            wk, bk = f"up_paths.0.{j}.up.0.weight", f"up_paths.0.{j}.up.0.bias"
            wp = self._pack(Pw[wk], L.PK_CT if zd[i] == 2 else L.PK_CT4, Cl, Cup, False)
            utiles = lib.bpx_convT3d_stats_tiles(Dl, Hl, Wl, zd[i])
            upart = _Stats.alloc(B, utiles, Cup, dev)
            up_raw = buf(i, Cup)
            L.check(lib.bpx_convT3d_k2s2_fwd(self.dt, B, Dl, Hl, Wl, zd[i], L.tview(dec_in), wp.data_ptr(), Pw[bk].data_ptr(), L.tview(up_raw),
                                             upart.data_ptr(), st))
            vox = S[i][0] * S[i][1] * S[i][2]
            urec = _recs(B, Cup, dev)
            _Stats.finalize(upart, B, utiles, Cup, vox, Pw[f"up_paths.0.{j}.up.1.weight"], Pw[f"up_paths.0.{j}.up.1.bias"], urec, Cup, 0, st)
            L.check(lib.bpx_norm_act_fwd(self.dt, B, vox, L.tview(up_raw), urec.data_ptr(), self.act, L.tview(cat[i], 0, Cup), st))
            out = buf(i, fm[i])
            blocks.append(self._conv_block_fwd(Pw, f"up_paths.0.{j}.conv_block", B, S[i], 2 * fm[i], fm[i], cat[i], None, L.tview(out), st, dev))
            ups.append((j, i, dec_in, up_raw, urec))
            dec_in = out
        # ---------------- heads ----------------------------------------------------------------------
        n_out = sum(cfg.out_channels)
        hw = torch.cat([Pw[f"heads.{h}.weight"].reshape(-1, fm[0]) for h in range(len(cfg.out_channels))], 0).contiguous()
        hb = torch.cat([Pw[f"heads.{h}.bias"] for h in range(len(cfg.out_channels))], 0).contiguous()
        vox0 = D0 * H0 * W0
        logits = torch.empty((B, n_out, D0, H0, W0), dtype=torch.float32, device=dev)
        L.check(lib.bpx_head_fwd(self.dt, vox0, B, L.tview(dec_in), hw.data_ptr(), hb.data_ptr(), n_out, head_act, logits.data_ptr(),
                                 n_out * vox0, vox0, st))
        if self.ndim == 2:
            logits = logits.reshape(B, n_out, H0, W0)
        ctx = dict(B=B, S=S, zd=zd, img=img, blocks=blocks, cat=cat, ups=ups, feat=dec_in, hw=hw, Pw=Pw) if save else None
        return logits, ctx

    # ---- backward -------------------------------------------------------------------------------------------------------
    def _norm_act_bwd(self, B, vox, C, dA, raw, rec, gamma, dgamma, dbeta, st, dev):
        """d(act(IN(raw))) -> d(raw): elementwise product with act' + the two InstanceNorm reductions, finalize, apply."""
        # (mixed mode, compute_dtype float16: `raw` is the forward pass's fp16 tensor, every gradient tensor bf16 - the codes of engine.ResUNetEngine:
        #  gdt = kernels that touch gradient tensors only, bdt = backward kernels that also read a forward activation)
        tiles = lib.bpx_norm_act_tiles(self.gdt, vox, C)
        red = torch.empty((B, tiles, 2, C), dtype=torch.float32, device=dev)
        g = torch.empty(raw.shape, dtype=self.gdtype, device=dev)
        L.check(lib.bpx_norm_act_bwd(self.bdt, B, vox, dA, L.tview(raw), rec.data_ptr(), self.act, L.NULL_T, L.tview(g), red.data_ptr(), st))
        coef = torch.empty((B, C, 4), dtype=torch.float32, device=dev)
        L.check(lib.bpx_norm_bwd_finalize(red.data_ptr(), B, tiles, C, vox, rec.data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), C,
                                          coef.data_ptr(), st))
        L.check(lib.bpx_norm_bwd_apply(self.bdt, B, vox, L.tview(g), L.tview(raw), coef.data_ptr(), L.NULL_T, L.tview(g), st))
        return g

    def _conv_block_bwd(self, Pw, G, cb: _CB, B, dA: "L.Tensor", img, dx_out: Optional["L.Tensor"], st, dev):
        D, H, W = cb.S
        vox = D * H * W
        p, C1 = cb.prefix, cb.cout
        k = lambda c, leaf: f"{p}.block.{c}.block.{leaf}"   # noqa: E731
        # output norm+act of the last conv
        g2 = self._norm_act_bwd(B, vox, C1, dA, cb.h[1], cb.rec[1], Pw[k(1, "1.weight")], G[k(1, "1.weight")], G[k(1, "1.bias")], st, dev)
        self._keep.append(g2)
        # conv 2: weights, then the input gradient fused with act' and the reductions of the first norm
        self._wgrad(B, cb.S, L.tview(cb.h[0]), cb.rec[0], self.act, L.tview(g2), 3, G[k(1, "0.weight")], G[k(1, "0.bias")], st, dev)
        g1 = torch.empty((B, D, H, W, C1), dtype=self.gdtype, device=dev)
        self._keep.append(g1)
        tiles = lib.bpx_conv3d_stats_tiles(self.dt, B, D, H, W, C1)
        red = torch.empty((B, tiles, 2, C1), dtype=torch.float32, device=dev)
        w2t = self._pack(Pw[k(1, "0.weight")], L.PK_K3_T, C1, C1, False)
        L.check(lib.bpx_conv3d_dgrad(self.bdt, B, D, H, W, L.tview(g2), w2t.data_ptr(), L.tview(cb.h[0]), cb.rec[0].data_ptr(), self.act,
                                     L.tview(g1), red.data_ptr(), st))
        coef = torch.empty((B, C1, 4), dtype=torch.float32, device=dev)
        L.check(lib.bpx_norm_bwd_finalize(red.data_ptr(), B, tiles, C1, vox, cb.rec[0].data_ptr(), Pw[k(0, "1.weight")].data_ptr(),
                                          G[k(0, "1.weight")].data_ptr(), G[k(0, "1.bias")].data_ptr(), C1, coef.data_ptr(), st))
        L.check(lib.bpx_norm_bwd_apply(self.bdt, B, vox, L.tview(g1), L.tview(cb.h[0]), coef.data_ptr(), L.NULL_T, L.tview(g1), st))
        # conv 1
        if img is not None:
            wsc = self._workspace(lib.bpx_conv3d_c1_wgrad_workspace(C1), dev)
            L.check(lib.bpx_conv3d_c1_wgrad(self.gdt, B, D, H, W, img.data_ptr(), L.tview(g1), G[k(0, "0.weight")].data_ptr(),
                                            G[k(0, "0.bias")].data_ptr(), wsc.data_ptr(), wsc.numel(), st))
            return
        self._wgrad(B, cb.S, L.tview(cb.x), None, 0, L.tview(g1), 3, G[k(0, "0.weight")], G[k(0, "0.bias")], st, dev)
        if dx_out is not None:
            w1t = self._pack(Pw[k(0, "0.weight")], L.PK_K3_T, cb.cin, C1, False)
            L.check(lib.bpx_conv3d_dgrad(self.gdt, B, D, H, W, L.tview(g1), w1t.data_ptr(), L.NULL_T, None, 0, dx_out, None, st))

    def backward(self, P: Dict[str, torch.Tensor], ctx, dlogits: torch.Tensor) -> Dict[str, torch.Tensor]:
        cfg = self.cfg
        B, S, zd, img, Pw = ctx["B"], ctx["S"], ctx["zd"], ctx["img"], ctx["Pw"]
        blocks: List[_CB] = ctx["blocks"]
        cat, ups, feat = ctx["cat"], ctx["ups"], ctx["feat"]
        fm, Lv = list(cfg.feature_maps), cfg.depth
        dev, st, T = dlogits.device, L.stream_ptr(), self.gdtype       # T: storage type of the gradient tensors
        self._keep = []
        # parameter gradients in the LIFTED shapes (one zero-filled slab: the wgrad kernels accumulate), un-lifted at the end
        names = list(Pw.keys())
        flat = torch.zeros(sum(Pw[n].numel() for n in names), dtype=torch.float32, device=dev)
        G: Dict[str, torch.Tensor] = {}
        o = 0
        for n in names:
            G[n] = flat[o:o + Pw[n].numel()].view(Pw[n].shape)
            o += Pw[n].numel()
        # ---- head -------------------------------------------------------------------------------
        n_out = sum(cfg.out_channels)
        D0, H0, W0 = S[0]
        vox0 = D0 * H0 * W0
        dl = dlogits.contiguous().float()
        dfeat = torch.empty((B, D0, H0, W0, fm[0]), dtype=T, device=dev)
        hwg = torch.zeros((n_out, fm[0]), dtype=torch.float32, device=dev)
        hbg = torch.zeros((n_out,), dtype=torch.float32, device=dev)
        hws = self._workspace(lib.bpx_head_bwd_workspace(fm[0], n_out), dev)
        L.check(lib.bpx_head_bwd(self.bdt, vox0, B, L.tview(feat), ctx["hw"].data_ptr(), n_out, dl.data_ptr(), n_out * vox0, vox0,
                                 L.tview(dfeat), hwg.data_ptr(), hbg.data_ptr(), hws.data_ptr(), hws.numel(), st))
        o = 0
        for h, oc in enumerate(cfg.out_channels):
            G[f"heads.{h}.weight"].copy_(hwg[o:o + oc].view(G[f"heads.{h}.weight"].shape))
            G[f"heads.{h}.bias"].copy_(hbg[o:o + oc])
            o += oc
        # ---- decoder: walk from level 0 down to the bottleneck ---------------------------------------------------------
        dcat: List[Optional[torch.Tensor]] = [None] * Lv
        dA = L.tview(dfeat)
        keep = [dfeat]
        for j, i, x_low, up_raw, urec in reversed(ups):
            Cl, Cup = fm[i + 1], fm[i]
            dcat[i] = torch.empty((B,) + S[i] + (2 * fm[i],), dtype=T, device=dev)
            self._conv_block_bwd(Pw, G, blocks[Lv + 1 + j], B, dA, None, L.tview(dcat[i]), st, dev)
            vox = S[i][0] * S[i][1] * S[i][2]
            pre = f"up_paths.0.{j}.up"
            dup = self._norm_act_bwd(B, vox, Cup, L.tview(dcat[i], 0, Cup), up_raw, urec, Pw[f"{pre}.1.weight"], G[f"{pre}.1.weight"],
                                     G[f"{pre}.1.bias"], st, dev)
            keep.append(dup)
            Dl, Hl, Wl = S[i + 1]
            ws = self._workspace(lib.bpx_convT3d_k2s2_wgrad_workspace(B, Dl, Hl, Wl, zd[i], Cl, Cup), dev)
            L.check(lib.bpx_convT3d_k2s2_wgrad(self.bdt, B, Dl, Hl, Wl, zd[i], L.tview(x_low), L.tview(dup), G[f"{pre}.0.weight"].data_ptr(),
                                               G[f"{pre}.0.bias"].data_ptr(), ws.data_ptr(), ws.numel(), st))
            dlow = torch.empty((B, Dl, Hl, Wl, Cl), dtype=T, device=dev)
            wt = self._pack(Pw[f"{pre}.0.weight"], L.PK_CT_T if zd[i] == 2 else L.PK_CT4_T, Cl, Cup, False)
            L.check(lib.bpx_convT3d_k2s2_dgrad(self.gdt, B, Dl, Hl, Wl, zd[i], L.tview(dup), wt.data_ptr(), L.tview(dlow), st))
            dA = L.tview(dlow)
            keep.append(dlow)
        # ---- bottleneck ---------------------------------------------------------------------------------------------
        dP = torch.empty((B,) + S[Lv] + (fm[Lv - 1],), dtype=T, device=dev)
        self._conv_block_bwd(Pw, G, blocks[Lv], B, dA, None, L.tview(dP), st, dev)
        # ---- encoder ------------------------------------------------------------------------------------------------
        for i in range(Lv - 1, -1, -1):
            D, H, W = S[i]
            # d(skip output) = d(concat)[..., Cup:] + unpool(dP), in place over the skip slice
            skipv = L.tview(dcat[i], fm[i], fm[i])
            L.check(lib.bpx_maxpool3d_bwd(self.bdt, B, D, H, W, zd[i], L.tview(cat[i], fm[i], fm[i]), L.tview(dP), skipv, skipv, st))
            if i > 0:
                dPn = torch.empty((B,) + S[i] + (fm[i - 1],), dtype=T, device=dev)
                self._conv_block_bwd(Pw, G, blocks[i], B, skipv, None, L.tview(dPn), st, dev)
                keep.append(dP)
                dP = dPn
            else:
                self._conv_block_bwd(Pw, G, blocks[0], B, skipv, img, None, st, dev)   # the image needs no gradient
        self._keep = []
        return unlift_grads(G, P)   # gradients in the shapes of the module's parameters
