"""Drop-in for ``biapy.models.rcan.rcan`` in 3-D (cfg 5): the trunk, and (round 3) the x``scale`` up-scaling stage.

Same constructor keywords as the reference (biapy/models/rcan.py:241-300), same ``state_dict`` keys and shapes (``sf``,
``rgs.g.module.r.module.{0,2}`` convolutions, ``rgs.g.module.r.module.3.module.{1,3}`` channel attention, ``rgs.g.module.n``
group tail, ``conv1``, ``conv2``), ordinary ``nn.Parameter``s; ``forward`` hands them to :class:`biapy_amd.rcan_engine.RCANEngine`.

``upscaling_layer=True`` in 3-D: the reference builds ``conv(filters, filters * scale**2) + nn.PixelShuffle(scale)`` (rcan.py:317-319), which is
2-D only - ``nn.PixelShuffle`` on 5-D tensors raises - so there is no reference behaviour to reproduce.  This class DEFINES the 3-D form as
the same rule with one more axis: ``upscale.0 = Conv3d(filters, filters * scale**3, 3)`` followed by a 3-D pixel shuffle
(out[n, c, s z + a, s y + b, s x + e] = conv[n, c s^3 + (a s + b) s + e, z, y, x]; oracle/rcan_oracle.py::pixel_shuffle3d), fused into the store
of the convolution (``bpx_conv3d_fwd_shuffle``: the 16 s^3-channel tensor never exists), filters = 16, scale 2..4, patches >= 32^3, 16-bit
storage.  Training (round 4): the stage's backward is the shuffle's adjoint (sub-positions gathered back into 16-channel blocks) followed by the
convolution's own weight / input gradient kernels per block on the low-resolution grid (``RCANEngine.backward``).  Parity: unpinned against BiaPy by
construction, outputs and every gradient checked against the oracle's restatement of the defined semantics (autograd through ``pixel_shuffle3d``).

Not covered (``NotImplementedError`` at construction): 2D, more than one input channel, filters other than 16 / 32, more than 4 output
channels.
Training goes through the linear output (``head_activations=["linear"]``, what the SR workflows use); other head activations
are inference-only (``predict``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .rcan_engine import RCANEngine
from .resunet import _ResUNetFn


class ChannelAttention(nn.Module):
    def __init__(self, num_features: int, reduction: int):
        super().__init__()
        self.module = nn.Sequential(nn.AdaptiveAvgPool3d(1), nn.Conv3d(num_features, num_features // reduction, kernel_size=1), nn.SiLU(inplace=True),
                                    nn.Conv3d(num_features // reduction, num_features, kernel_size=1), nn.Sigmoid())


class RCAB_rcan(nn.Module):
    def __init__(self, num_features: int, reduction: int):
        super().__init__()
        self.module = nn.Sequential(nn.Conv3d(num_features, num_features, kernel_size=3, padding="same"), nn.SiLU(inplace=True),
                                    nn.Conv3d(num_features, num_features, kernel_size=3, padding="same"), ChannelAttention(num_features, reduction))


class RG(nn.Module):
    def __init__(self, num_features: int, num_rcab: int, reduction: int):
        super().__init__()
        self.module = nn.Sequential(*([RCAB_rcan(num_features, reduction) for _ in range(num_rcab)] +
                                      [nn.Conv3d(num_features, num_features, kernel_size=3, padding="same")]))


class rcan(nn.Module):
    _bpx_dropin = True   # train_engine: the training-time model_call_func of this class is to_pytorch_format -> forward
    _HEAD = {"linear": 0, "sigmoid": 1, "tanh": 2}

    def __init__(self, ndim, num_channels=3, filters=64, scale=2, num_rg=10, num_rcab=20, reduction=16, upscaling_layer=True,
                 out_channels: Optional[int] = None, head_activations: Optional[List[str]] = None, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        if type(scale) is not int and isinstance(scale, Sequence):
            scale = scale[0]
        if ndim != 3:
            raise NotImplementedError("biapy_amd.rcan: the 3-D network is what runs on the MI355X path")
        if upscaling_layer and (filters != 16 or int(scale) not in (2, 3, 4)):
            raise NotImplementedError("biapy_amd.rcan: the up-scaling stage takes 16 filters and scale 2, 3 or 4 (3-D pixel shuffle fused into the conv store)")
        if out_channels is None:
            out_channels = num_channels
        self.ndim, self.upscaling_layer = ndim, upscaling_layer
        act_name = (head_activations[0] if head_activations else "linear").lower().removeprefix("ce_")
        if act_name not in self._HEAD:
            raise NotImplementedError(f"biapy_amd.rcan: output activation {act_name!r}")
        self.head_code = self._HEAD[act_name]
        self.scale = int(scale) if upscaling_layer else 0
        self.cfg = dict(num_channels=num_channels, filters=filters, num_rg=num_rg, num_rcab=num_rcab, reduction=reduction, out_channels=out_channels,
                        scale=self.scale)
        self.compute_dtype = compute_dtype
        self._engine: Optional[RCANEngine] = None
        RCANEngine(dtype=compute_dtype, **self.cfg)                      # validates the configuration (raises NotImplementedError)
        self.sf = nn.Conv3d(num_channels, filters, kernel_size=3, padding="same")
        self.rgs = nn.Sequential(*[RG(filters, num_rcab, reduction) for _ in range(num_rg)])
        self.conv1 = nn.Conv3d(filters, filters, kernel_size=3, padding="same")
        if upscaling_layer:                      # 3-D form of rcan.py:317-319 (see the module docstring): s^3 sub-positions per feature channel
            self.upscale = nn.Sequential(nn.Conv3d(filters, filters * self.scale ** 3, kernel_size=3, padding="same"))
        self.conv2 = nn.Conv3d(filters, out_channels, kernel_size=3, padding="same")

    def engine(self) -> RCANEngine:
        if self._engine is None or self._engine.dtype != self.compute_dtype:
            self._engine = RCANEngine(dtype=self.compute_dtype, **self.cfg)
        return self._engine

    supported_compute_dtypes = (torch.float32, torch.bfloat16, torch.float16)   # float16: fp16 forward / activations; training = the mixed mode (bf16 gradients, round 4)

    def forward(self, x) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("biapy_amd.rcan runs on the MI355X only (input is on %s); there is no CPU path" % x.device)
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        x = x.to(torch.float32)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            if self.head_code != 0:
                raise NotImplementedError("biapy_amd.rcan trains through the linear output; use head_activations=['linear']")
            return _ResUNetFn.apply(x, self.engine(), names, *params)
        P = {n: p.detach() for n, p in zip(names, params)}
        y, _ = self.engine().forward(P, x, head_act=self.head_code, save=False)
        return y
