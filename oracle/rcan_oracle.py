"""CPU oracle of the RCAN trunk (no up-scaling layer), plain PyTorch CPU fp32.  TEST INFRASTRUCTURE ONLY (tests/ only).

Parity status: PINNED for the trunk; the 3-D up-scaling stage (``upscale`` key present) is **parity unpinned** - the reference's own
3-D branch raises, the semantics are DEFINED by ``pixel_shuffle3d`` below (the extension of ``nn.PixelShuffle`` to three axes)
and checked against ``torch.nn.functional.pixel_shuffle`` in the 2-D limit.  For the trunk: - ``tests/golden/make_golden.py rcan`` builds the reference ``rcan(ndim=3,
upscaling_layer=False)`` in the build container and commits weights, input, output, an L1 loss and gradients
(``tests/golden/rcan_golden.npz``).  The reference's 3-D up-scaling branch (``nn.PixelShuffle`` on 5-D tensors) raises, so
there is nothing to pin for it.

Restates (paths relative to /root/reference): biapy/models/rcan.py - ``ChannelAttention`` (avg-pool -> 1x1 conv -> SiLU ->
1x1 conv -> sigmoid, ``x * module(x)``), ``RCAB_rcan`` (``x + [conv, SiLU, conv, ChannelAttention](x)``), ``RG``
(``x + [RCAB x n, conv](x)``), ``rcan.forward`` (``sf`` -> groups -> ``conv1`` -> ``+ residual`` -> ``conv2`` -> head activation).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def _conv(x, sd, key):
    w = sd[key + ".weight"]
    return F.conv3d(x, w, sd[key + ".bias"], padding=w.shape[-1] // 2)


def rcab(x, sd, p):
    h = _conv(F.silu(_conv(x, sd, f"{p}.module.0")), sd, f"{p}.module.2")
    s = torch.sigmoid(_conv(F.silu(_conv(h.mean((2, 3, 4), keepdim=True), sd, f"{p}.module.3.module.1")), sd, f"{p}.module.3.module.3"))
    return x + h * s


def pixel_shuffle3d(x: torch.Tensor, s: int) -> torch.Tensor:
    """(B, C s^3, Z, Y, X) -> (B, C, sZ, sY, sX): out[n, c, s z + a, s y + b, s x + e] = x[n, c s^3 + (a s + b) s + e, z, y, x].
    ``nn.PixelShuffle`` (rcan.py:317-319 uses it for the 2-D network; on 5-D tensors it raises) maps channel c s^2 + a s + b to the
    sub-position (a, b) of channel c; this is the same rule with one more axis - for Z = 1 and the a = 0 channels it IS
    ``F.pixel_shuffle`` (tests/test_oracle_golden.py)."""
    B, Cs, Z, Y, X = x.shape
    C = Cs // (s ** 3)
    x = x.reshape(B, C, s, s, s, Z, Y, X)
    return x.permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, C, Z * s, Y * s, X * s)


def rcan_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, num_rg: int, num_rcab: int, head_activation: str = "linear", scale: int = 0) -> torch.Tensor:
    x = _conv(x, sd, "sf")
    residual = x
    for g in range(num_rg):
        z = x
        for r in range(num_rcab):
            z = rcab(z, sd, f"rgs.{g}.module.{r}")
        x = x + _conv(z, sd, f"rgs.{g}.module.{num_rcab}")
    x = _conv(x, sd, "conv1") + residual
    if scale:                                           # rcan.py:344-345 `x = self.upscale(x)`: conv(filters -> filters * s^3) + 3-D pixel shuffle
        x = pixel_shuffle3d(_conv(x, sd, "upscale.0"), scale)
    x = _conv(x, sd, "conv2")
    return {"linear": lambda t: t, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[head_activation](x)
