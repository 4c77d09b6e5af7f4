"""CPU oracle of ResUNet++ (3D), plain PyTorch CPU fp32.  TEST INFRASTRUCTURE ONLY (tests/ only).

Prepared for the row that is still open on the device (SURVEY.md row X / cfg 4; DESIGN.md "Plan for the row that is still open").
Parity status: PINNED - ``tests/golden/make_golden.py resunetpp`` builds the reference ``ResUNetPlusPlus`` in the build container and
commits weights, input, logits, loss and gradient norms (``tests/golden/resunetpp_golden.npz``).

Restates (paths relative to /root/reference):
  * graph ............. biapy/models/resunet++.py:435-466 (encoder with SE and pooling from the second level on, ASPP bridge,
                        attention + ResUpBlock decoder, ASPP output block, heads)
  * residual block .... biapy/models/blocks.py:1304-1378, :1456-1459 with ``skip_k_size = 3`` and ``skip_norm`` (shortcut = conv3 + IN)
  * SE ................ biapy/models/blocks.py:1119-1191 (avg-pool -> Linear -> ReLU -> Linear -> sigmoid, no biases)
  * ASPP .............. biapy/models/heads.py:13-133 (dilated conv -> ReLU -> IN, rates 6 / 12 / 18, concat, 1x1 conv)
  * attention ......... biapy/models/blocks.py:2168-2298
  * up block .......... biapy/models/blocks.py:1603-1655 (ConvTranspose(in -> in), cat([up, bridge]), residual block)
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .net_oracle import _act


def _in(x, sd, key):
    return F.instance_norm(x, None, None, sd[key + ".weight"], sd[key + ".bias"], True, 0.1, 1e-5)


def _conv(x, sd, key, dilation=1):
    w = sd[key + ".weight"]
    k = w.shape[-1]
    return F.conv3d(x, w, sd.get(key + ".bias"), padding=dilation * (k // 2), dilation=dilation)


def res_block(x, sd, p, first: bool, act: str):
    h = x
    i = 0
    if not first:
        h = _act(_in(h, sd, f"{p}.block.0"), act)
        i = 2
    h = _act(_in(_conv(h, sd, f"{p}.block.{i}.block.0"), sd, f"{p}.block.{i}.block.1"), act)
    h = _conv(h, sd, f"{p}.block.{i + 1}.block.0")
    return h + _in(_conv(x, sd, f"{p}.shortcut.0"), sd, f"{p}.shortcut.1")


def sqex(x, sd, p):
    y = x.mean((2, 3, 4))
    y = torch.sigmoid(F.relu(y @ sd[f"{p}.excitation.0.weight"].t()) @ sd[f"{p}.excitation.2.weight"].t())
    return x * y[:, :, None, None, None]


def aspp(x, sd, p, rates=(6, 12, 18)):
    outs = [_in(F.relu(_conv(x, sd, f"{p}.aspp_block{j + 1}.0", dilation=r)), sd, f"{p}.aspp_block{j + 1}.2") for j, r in enumerate(rates)]
    return _conv(torch.cat(outs, 1), sd, f"{p}.output")


def attention(x1, x2, sd, p, pool):
    e = F.max_pool3d(_conv(F.relu(_in(x1, sd, f"{p}.conv_encoder.0")), sd, f"{p}.conv_encoder.2"), pool)
    d = _conv(F.relu(_in(x2, sd, f"{p}.conv_decoder.0")), sd, f"{p}.conv_decoder.2")
    a = _conv(F.relu(_in(e + d, sd, f"{p}.conv_attn.0")), sd, f"{p}.conv_attn.2")
    return a * x2


def resunetpp_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, feature_maps: Sequence[int], z_down: Optional[Sequence[int]] = None,
                      activation: str = "elu", n_heads: int = 1) -> torch.Tensor:
    depth = len(feature_maps) - 2
    zd = list(z_down) if z_down is not None else [2] * (depth + 1)
    blocks = []
    for i in range(depth + 1):
        x = res_block(x, sd, f"down_path.{i}", i == 0, activation)
        if i < depth:
            x = sqex(x, sd, f"sqex_blocks.{i}")
        if i != 0:
            x = F.max_pool3d(x, (zd[i], 2, 2))
        blocks.append(x)
    x = aspp(x, sd, "aspp_bridge")
    for j in range(depth):
        i = depth - 1 - j
        x = attention(blocks[-j - 2], x, sd, f"attentions.0.{j}", (zd[i + 1], 2, 2))
        up = F.conv_transpose3d(x, sd[f"up_paths.0.{j}.up.weight"], sd[f"up_paths.0.{j}.up.bias"], stride=(zd[i + 1], 2, 2))
        x = res_block(torch.cat([up, blocks[-j - 2]], 1), sd, f"up_paths.0.{j}.conv_block", False, activation)
    x = aspp(x, sd, "aspp_out.0")
    return torch.cat([_conv(x, sd, f"heads.{h}") for h in range(n_heads)], 1)
