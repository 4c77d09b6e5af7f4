"""CPU oracle for the plain (non-residual) 2D / 3D U-Net forward (plain PyTorch CPU, fp32).

TEST INFRASTRUCTURE ONLY (imported by ``tests/`` only).  Nothing under ``biapy_amd/`` imports it.

Parity status: PINNED against the imported reference (``tests/golden/make_golden.py unet`` builds the reference
``U_Net`` in the build container, 2D and 3D, runs it on seeded inputs and commits weights, inputs, logits, loss and
gradient norms under ``tests/golden/unet_golden.npz``; ``tests/test_oracle_golden.py`` checks this restatement
against them).

Reference being restated (paths relative to /root/reference):
  * network graph ....... biapy/models/unet.py:382-394 (encoder / bottleneck / decoder walk), :396-420 (heads)
  * conv block .......... biapy/models/blocks.py:120-167  (nconvs x [Conv -> InstanceNorm -> act])
  * up block ............ biapy/models/blocks.py:602-614, :656-668 (ConvTranspose(in->out) -> norm -> act; cat([up, bridge]))
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .net_oracle import _act


def _conv_nd(x, w, b):
    pad = tuple(k // 2 for k in w.shape[2:])
    return (F.conv2d if w.dim() == 4 else F.conv3d)(x, w, b, padding=pad)


def _in(x, w, b):
    return F.instance_norm(x, None, None, w, b, True, 0.1, 1e-5)


def conv_block(x, sd, prefix: str, nconvs: int, act: str):
    for c in range(nconvs):
        x = _conv_nd(x, sd[f"{prefix}.block.{c}.block.0.weight"], sd[f"{prefix}.block.{c}.block.0.bias"])
        x = _act(_in(x, sd[f"{prefix}.block.{c}.block.1.weight"], sd[f"{prefix}.block.{c}.block.1.bias"]), act)
    return x


def unet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, feature_maps: Sequence[int], z_down: Optional[Sequence[int]] = None,
                 activation: str = "elu", nconvs: int = 2, n_heads: int = 1) -> torch.Tensor:
    """x: (B,C,Y,X) or (B,C,Z,Y,X) fp32 -> logits.  ``len(feature_maps) - 1`` pooling levels (unet.py:188 depth)."""
    nd = x.dim() - 2
    depth = len(feature_maps) - 1
    zd = list(z_down) if z_down is not None else [2] * depth
    skips = []
    for i in range(depth):
        x = conv_block(x, sd, f"down_path.{i}", nconvs, activation)
        skips.append(x)
        x = F.max_pool2d(x, 2) if nd == 2 else F.max_pool3d(x, (zd[i], 2, 2))
    x = conv_block(x, sd, "bottleneck", nconvs, activation)
    for j, i in enumerate(range(depth - 1, -1, -1)):
        w, b = sd[f"up_paths.0.{j}.up.0.weight"], sd[f"up_paths.0.{j}.up.0.bias"]
        up = F.conv_transpose2d(x, w, b, stride=2) if nd == 2 else F.conv_transpose3d(x, w, b, stride=(zd[i], 2, 2))
        up = _act(_in(up, sd[f"up_paths.0.{j}.up.1.weight"], sd[f"up_paths.0.{j}.up.1.bias"]), activation)
        x = conv_block(torch.cat([up, skips[i]], 1), sd, f"up_paths.0.{j}.conv_block", nconvs, activation)
    return torch.cat([_conv_nd(x, sd[f"heads.{h}.weight"], sd[f"heads.{h}.bias"]) for h in range(n_heads)], 1)
