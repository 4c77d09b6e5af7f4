"""CPU oracle of the RCAN trunk (no up-scaling layer), plain PyTorch CPU fp32.  TEST INFRASTRUCTURE ONLY (tests/ only).

Parity status: PINNED for the trunk - ``tests/golden/make_golden.py rcan`` builds the reference ``rcan(ndim=3,
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


def rcan_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, num_rg: int, num_rcab: int, head_activation: str = "linear") -> torch.Tensor:
    x = _conv(x, sd, "sf")
    residual = x
    for g in range(num_rg):
        z = x
        for r in range(num_rcab):
            z = rcab(z, sd, f"rgs.{g}.module.{r}")
        x = x + _conv(z, sd, f"rgs.{g}.module.{num_rcab}")
    x = _conv(x, sd, "conv1") + residual
    x = _conv(x, sd, "conv2")
    return {"linear": lambda t: t, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[head_activation](x)
