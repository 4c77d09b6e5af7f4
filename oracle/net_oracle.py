"""CPU oracle for the 3D residual U-Net forward / train step (plain PyTorch CPU, fp32).

TEST INFRASTRUCTURE ONLY.  Used as the checker by ``tests/``, ``__graft_entry__.smoke()`` and as
the ``cpu_baseline`` ("port") leg of ``bench.py``.  Nothing under ``biapy_amd/`` imports it.

Parity status: PINNED against the imported reference (``tests/golden/make_golden.py`` builds the
reference ``ResUNet`` in the build container, runs it on seeded inputs and commits inputs,
weights and outputs under ``tests/golden/resunet_*.npz``; ``tests/test_oracle_golden.py`` checks
this restatement against them, tolerance 2e-5 abs on logits - the arithmetic underneath is
PyTorch's own CPU conv/norm kernels in both cases).

The restatement is functional: it walks a ``state_dict`` with the reference's key names
 instead of instantiating modules.

Reference being restated (paths relative to /root/reference):
  * network graph ....... biapy/models/resunet.py:352-446 (forward), :238-304 (construction)
  * residual block ...... biapy/models/blocks.py:1304-1378, :1456-1459
  * up block ............ biapy/models/blocks.py:1603-1655
  * conv block .......... biapy/models/blocks.py:146-167
  * norm / activation ... biapy/models/blocks.py:2113-2127 (InstanceNorm3d affine, eps 1e-5), :1986-1998
  * init ................ biapy/models/blocks.py:2301-2336 (Xavier-uniform on Conv3d, zero bias;
                          ConvTranspose3d keeps the PyTorch default)
  * loss ................ biapy/engine/metrics.py:543-544, :575-576 (BCEWithLogitsLoss, mean)
  * head activation ..... biapy/engine/base_workflow.py:1403-1457 (sigmoid for ce_sigmoid)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


def _act(x: torch.Tensor, name: str) -> torch.Tensor:
    if name == "elu":
        return F.elu(x, alpha=1.0)
    if name == "relu":
        return F.relu(x)
    if name == "silu":
        return F.silu(x)
    if name == "leaky_relu":
        return F.leaky_relu(x)                      # nn.LeakyReLU(): slope 0.01
    if name == "gelu":
        return F.gelu(x)                            # nn.GELU(): exact erf form
    if name == "tanh":
        return torch.tanh(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "softplus":
        return F.softplus(x)                        # beta 1, threshold 20
    if name in ("none", "linear"):
        return x
    raise ValueError(name)


def _norm(x: torch.Tensor, sd, key: str, kind: str, groups: int = 8) -> torch.Tensor:
    if kind == "none":
        return x
    w, b = sd[key + ".weight"], sd[key + ".bias"]
    if kind == "in":
        return F.instance_norm(x, None, None, w, b, True, 0.1, 1e-5)
    if kind == "gn":  # what the reference's "gn" intends: GroupNorm(8, C) in 3D (SURVEY.md headline facts)
        return F.group_norm(x, groups, w, b, 1e-5)
    raise ValueError(kind)


def _conv(x, sd, key):
    w = sd[key + ".weight"]
    pad = tuple(k // 2 for k in w.shape[2:])      # "same" for the odd kernels of the path: 3, 1, (1,3,3), (3,3)
    return (F.conv2d if w.dim() == 4 else F.conv3d)(x, w, sd.get(key + ".bias"), padding=pad)


def res_conv_block(x, sd, prefix: str, first_block: bool, act: str, norm: str, drop=None):
    """block(x) + shortcut(x) for the default ``conv_norm_act`` order with two convolutions.
    ``drop`` = (p, keep mask): ``nn.Dropout(p)`` of the first ConvBlock (blocks.py:163: after Conv -> Norm -> Act) with its mask made explicit."""
    h = x
    i = 0
    if not first_block:
        if norm != "none":
            h = _norm(h, sd, f"{prefix}.block.0", norm)
            i = 1
        h = _act(h, act)
        i += 1
    h = _conv(h, sd, f"{prefix}.block.{i}.block.0")
    h = _norm(h, sd, f"{prefix}.block.{i}.block.1", norm)
    h = _act(h, act)
    if drop is not None:
        p, keep = drop
        h = h * keep.to(h.dtype) / (1.0 - p)
    h = _conv(h, sd, f"{prefix}.block.{i + 1}.block.0")
    return h + _conv(x, sd, f"{prefix}.shortcut.0")


def resunet_forward(
    sd: Dict[str, torch.Tensor],
    x: torch.Tensor,
    feature_maps: Sequence[int],
    z_down: Optional[Sequence[int]] = None,
    yx_down: Optional[Sequence[int]] = None,
    activation: str = "elu",
    normalization: str = "in",
    n_heads: int = 1,
    dropout: Optional[Dict[str, tuple]] = None,
) -> torch.Tensor:
    """x: (B,C,Z,Y,X) fp32 -> logits (B,sum(out_ch),Z,Y,X).
    ``dropout``: {block prefix ("down_path.0", "bottleneck", "up_paths.0.1.conv_block", ...): (p, keep mask (B,C,Z,Y,X))} - training-mode dropout
    (resunet.py:250, :270, :299 hand ``drop_values`` to the blocks) with the masks made explicit; None = evaluation mode / p = 0."""
    dropout = dropout or {}
    depth = len(feature_maps) - 1
    z_down = list(z_down) if z_down is not None else [2] * depth
    yx_down = list(yx_down) if yx_down is not None else [2] * depth
    skips: List[torch.Tensor] = []
    two_d = x.dim() == 4                               # 2D network (resunet.py:195-207 picks the 2D layer classes)
    if "pre_upsampling.weight" in sd:                  # super-resolution, resunet.py:206-213 / :368-369: ConvTranspose(k = s = factor)
        w = sd["pre_upsampling.weight"]
        x = F.conv_transpose3d(x, w, sd["pre_upsampling.bias"], stride=tuple(w.shape[2:]))
    for i in range(depth):
        x = res_conv_block(x, sd, f"down_path.{i}", i == 0, activation, normalization, dropout.get(f"down_path.{i}"))
        skips.append(x)
        x = F.max_pool2d(x, yx_down[i]) if two_d else F.max_pool3d(x, (z_down[i], yx_down[i], yx_down[i]))
    x = res_conv_block(x, sd, "bottleneck", False, activation, normalization, dropout.get("bottleneck"))
    for j, i in enumerate(range(depth - 1, -1, -1)):
        s = (z_down[i], yx_down[i], yx_down[i])
        if two_d:
            up = F.conv_transpose2d(x, sd[f"up_paths.0.{j}.up.weight"], sd[f"up_paths.0.{j}.up.bias"], stride=yx_down[i])
        else:
            up = F.conv_transpose3d(x, sd[f"up_paths.0.{j}.up.weight"], sd[f"up_paths.0.{j}.up.bias"], stride=s)
        x = torch.cat([up, skips[i]], 1)
        x = res_conv_block(x, sd, f"up_paths.0.{j}.conv_block", False, activation, normalization, dropout.get(f"up_paths.0.{j}.conv_block"))
    if "post_upsampling.weight" in sd:                 # resunet.py:326-333 / :399-400
        w = sd["post_upsampling.weight"]
        x = F.conv_transpose3d(x, w, sd["post_upsampling.bias"], stride=tuple(w.shape[2:]))
    outs = [_conv(x, sd, f"heads.{h}") for h in range(n_heads)]
    return torch.cat(outs, 1)


def init_state_dict(
    in_ch: int,
    feature_maps: Sequence[int],
    out_channels: Sequence[int] = (1,),
    z_down: Optional[Sequence[int]] = None,
    yx_down: Optional[Sequence[int]] = None,
    normalization: str = "in",
    seed: int = 0,
) -> Dict[str, torch.Tensor]:
    """Random weights with the reference's key names, shapes and init *distributions*.

    This does NOT reproduce the reference's RNG stream (module construction order differs); it is
    for synthetic benchmarks and oracle-vs-HIP parity where both sides consume the same dict.
    """
    g = torch.Generator().manual_seed(seed)
    depth = len(feature_maps) - 1
    z_down = list(z_down) if z_down is not None else [2] * depth
    yx_down = list(yx_down) if yx_down is not None else [2] * depth
    sd: Dict[str, torch.Tensor] = {}

    def xavier(shape):
        rf = 1
        for s in shape[2:]:
            rf *= s
        bound = math.sqrt(6.0 / (shape[1] * rf + shape[0] * rf))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def conv(key, cin, cout, k):
        sd[key + ".weight"] = xavier((cout, cin, k, k, k))
        sd[key + ".bias"] = torch.zeros(cout)

    def norm(key, c):
        if normalization != "none":
            # perturb the affine so parity tests exercise gamma/beta (reference init is 1/0)
            sd[key + ".weight"] = 1 + 0.1 * (torch.rand(c, generator=g) * 2 - 1)
            sd[key + ".bias"] = 0.1 * (torch.rand(c, generator=g) * 2 - 1)

    def block(prefix, cin, cout, first):
        i = 0
        if not first:
            norm(f"{prefix}.block.0", cin)
            i = 2 if normalization != "none" else 1
        conv(f"{prefix}.block.{i}.block.0", cin, cout, 3)
        norm(f"{prefix}.block.{i}.block.1", cout)
        conv(f"{prefix}.block.{i + 1}.block.0", cout, cout, 3)
        conv(f"{prefix}.shortcut.0", cin, cout, 1)

    c = in_ch
    for i in range(depth):
        block(f"down_path.{i}", c, feature_maps[i], i == 0)
        c = feature_maps[i]
    block("bottleneck", c, feature_maps[-1], False)
    c = feature_maps[-1]
    for j, i in enumerate(range(depth - 1, -1, -1)):
        kz, kyx = z_down[i], yx_down[i]
        fan = c * kz * kyx * kyx  # PyTorch default ConvTranspose init: kaiming_uniform(a=sqrt(5)) on weight.size(1)*rf
        bound = 1.0 / math.sqrt(fan)
        sd[f"up_paths.0.{j}.up.weight"] = (torch.rand((c, c, kz, kyx, kyx), generator=g) * 2 - 1) * bound
        sd[f"up_paths.0.{j}.up.bias"] = (torch.rand(c, generator=g) * 2 - 1) * bound
        block(f"up_paths.0.{j}.conv_block", c + feature_maps[i], feature_maps[i], False)
        c = feature_maps[i]
    for h, oc in enumerate(out_channels):
        conv(f"heads.{h}", c, oc, 1)
    return sd


def bce_with_logits(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return F.binary_cross_entropy_with_logits(logits, target.to(torch.float32))


def train_step_grads(sd, x, target, **kw):
    """Returns (loss, logits, {param key: grad}) for one BCE-with-logits step."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = resunet_forward(params, x, **kw)
    loss = bce_with_logits(logits, target)
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.detach(), logits.detach(), dict(zip(params.keys(), grads))


def dice(prob: torch.Tensor, target: torch.Tensor, thr: float = 0.5) -> float:
    """Hard Dice of the binarised prediction vs a binary target (the parity metric, delta < 1e-4)."""
    p = (prob > thr).to(torch.float64)
    t = (target > 0.5).to(torch.float64)
    return float((2 * (p * t).sum() + 1e-5) / (p.sum() + t.sum() + 1e-5))


def count_flops_forward(in_ch, feature_maps, vol, z_down=None, yx_down=None, out_channels=(1,)) -> int:
    """2*MAC of every Conv3d / ConvTranspose3d for one sample (BASELINE.md section 3 convention)."""
    depth = len(feature_maps) - 1
    z_down = list(z_down) if z_down is not None else [2] * depth
    yx_down = list(yx_down) if yx_down is not None else [2] * depth
    D = list(vol)
    total = 0

    def vox():
        return D[0] * D[1] * D[2]

    def block(cin, cout):
        return 2 * vox() * (27 * cin * cout + 27 * cout * cout + cin * cout)

    c = in_ch
    for i in range(depth):
        total += block(c, feature_maps[i])
        c = feature_maps[i]
        D = [D[0] // z_down[i], D[1] // yx_down[i], D[2] // yx_down[i]]
    total += block(c, feature_maps[-1])
    c = feature_maps[-1]
    for i in range(depth - 1, -1, -1):
        k = z_down[i] * yx_down[i] * yx_down[i]
        total += 2 * vox() * c * c * k
        D = [D[0] * z_down[i], D[1] * yx_down[i], D[2] * yx_down[i]]
        total += block(c + feature_maps[i], feature_maps[i])
        c = feature_maps[i]
    total += 2 * vox() * c * sum(out_channels)
    return total
