"""Drop-in for ``biapy.models.unet.U_Net`` (2D and 3D) running on the MI355X engine (cfg 1 family).

Contract kept from the reference (biapy/models/unet.py:36-62 constructor, :366-444 forward; registry call at
biapy/models/__init__.py:120-145): same constructor keyword arguments, ``forward(x)`` on ``(B,C,Y,X)`` / ``(B,C,Z,Y,X)``
float32 returning the prediction tensor, identical ``state_dict()`` keys and shapes (checkpoints load with ``strict=True`` both
ways), ordinary ``nn.Parameter``s.  The module tree only owns the parameters under the reference's names; ``forward`` hands
them to :class:`biapy_amd.unet_engine.UNetEngine` and the gradients come from its hand-written backward through one
``torch.autograd.Function``.

Configurations the engine does not cover (normalisation other than "in", larger_io, separated decoders, contrastive head,
SR up-sampling, anisotropic (1,k,k) kernels in 3D, YX_DOWN != 2, Z_DOWN outside {1,2}, conv_layers != 2, pre-activation
order, dropout, "upsampling" up-mode) raise ``NotImplementedError`` at construction.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .engine import NetConfig
from .resunet import ResUNet, _act_layer, _ResUNetFn
from .unet_engine import UNetEngine


class ConvBlock(nn.Module):
    """Parameter holder named like blocks.py:120-167: ``block = Sequential(nconvs x ConvBlock(conv, norm, act))``."""

    def __init__(self, ndim: int, cin: int, cout: int, k: int, act: str, nconvs: int = 1):
        super().__init__()
        conv = nn.Conv2d if ndim == 2 else nn.Conv3d
        norm = nn.InstanceNorm2d if ndim == 2 else nn.InstanceNorm3d
        if nconvs > 1:
            self.block = nn.Sequential(*[ConvBlock(ndim, cin if i == 0 else cout, cout, k, act) for i in range(nconvs)])
        else:
            self.block = nn.Sequential(conv(cin, cout, kernel_size=k, padding="same"), norm(cout, affine=True, momentum=0.1), _act_layer(act))


class UpBlock(nn.Module):
    """Parameter holder named like blocks.py:510-668 (``up = Sequential(ConvTranspose(in->out), norm, act)``)."""

    def __init__(self, ndim: int, cin: int, cout: int, cbridge: int, k: int, act: str, z_down: int, nconvs: int):
        super().__init__()
        if ndim == 2:
            up = nn.ConvTranspose2d(cin, cout, kernel_size=(2, 2), stride=(2, 2))
            norm = nn.InstanceNorm2d(cout, affine=True, momentum=0.1)
        else:
            up = nn.ConvTranspose3d(cin, cout, kernel_size=(z_down, 2, 2), stride=(z_down, 2, 2))
            norm = nn.InstanceNorm3d(cout, affine=True, momentum=0.1)
        self.up = nn.Sequential(up, norm, _act_layer(act))
        self.attention_gate = None
        self.conv_block = ConvBlock(ndim, cout + cbridge, cout, k, act, nconvs)


class U_Net(nn.Module):
    _bpx_dropin = True   # train_engine: the training-time model_call_func of this class is to_pytorch_format -> forward
    # float16 (round 4) = the mixed training mode of the ResUNet: fp16 forward / activations, bf16 gradients (unet_engine.py uses the same dtype codes)
    supported_compute_dtypes = (torch.float32, torch.bfloat16, torch.float16)

    def __init__(
        self,
        image_shape=(256, 256, 1),
        activation="ELU",
        feature_maps=[32, 64, 128, 256],
        drop_values=[0.1, 0.1, 0.1, 0.1],
        normalization="none",
        k_size=3,
        upsample_layer="convtranspose",
        yx_down=[2, 2, 2, 2],
        z_down=[2, 2, 2, 2],
        output_channels=[1],
        separated_decoders=False,
        divide_decoder_feature_maps=False,
        output_channel_info=["F"],
        explicit_activations: bool = False,
        head_activations: List[str] = ["ce_sigmoid"],
        upsampling_factor=(),
        upsampling_position="pre",
        isotropy=False,
        larger_io=True,
        conv_layers: List[int] = [2, 2, 2, 2, 2],
        contrast: bool = False,
        contrast_proj_dim: int = 256,
        return_one_tensor: bool = False,
        conv_block_order: str = "conv_norm_act",
        compute_dtype: torch.dtype = torch.bfloat16,
    ):
        super().__init__()
        if len(output_channels) == 0:
            raise ValueError("'output_channels' needs to has at least one value")
        act = activation.lower()
        depth = len(feature_maps) - 1
        ndim = 3 if len(image_shape) == 4 else 2
        iso = [isotropy] * len(feature_maps) if isinstance(isotropy, bool) else list(isotropy)

        def unsupported(what):
            raise NotImplementedError(f"biapy_amd.U_Net: {what} is outside the MI355X hot path; use the reference PyTorch class for it")

        if k_size != 3 or (ndim == 3 and not all(iso)):
            unsupported("kernel size != 3 or anisotropic (1,k,k) kernels")
        if list(yx_down)[:depth] != [2] * depth:
            unsupported("YX_DOWN other than 2")
        if ndim == 3 and (len(list(z_down)) < depth or any(int(v) not in (1, 2) for v in list(z_down)[:depth])):
            unsupported("Z_DOWN other than 1 or 2")
        if upsample_layer != "convtranspose":
            unsupported("upsample_layer != 'convtranspose'")
        if separated_decoders or contrast or larger_io or len(upsampling_factor) > 0 or divide_decoder_feature_maps:
            unsupported("separated decoders / contrastive head / larger_io / super-resolution up-sampling")
        if conv_block_order != "conv_norm_act" or list(conv_layers)[: depth + 1] != [2] * (depth + 1):
            unsupported("conv_block_order != 'conv_norm_act' or conv_layers != 2")
        if any(float(d) > 0 for d in drop_values):
            unsupported("dropout")
        if explicit_activations or "class" in output_channel_info:
            unsupported("explicit head activations / classification head")
        self.depth, self.ndim = depth, ndim
        self.z_down, self.yx_down = z_down, yx_down
        self.output_channels = output_channels
        self.output_channel_info = output_channel_info
        self.head_activations = list(head_activations)
        self.return_class = self.contrast = self.explicit_activations = False
        self.return_one_tensor = return_one_tensor
        in_ch = image_shape[-1]
        zd = [int(v) for v in list(z_down)[:depth]] if ndim == 3 else [1] * depth
        self.cfg = NetConfig(in_ch=in_ch, feature_maps=list(feature_maps), out_channels=tuple(output_channels), activation=act,
                             normalization=normalization, z_down=zd)
        if self.cfg.true_feature_maps is not None:       # only the ResUNet engine runs widths that are not multiples of 16 (zero-padded); refuse here, at construction
            unsupported(f"feature_maps {list(feature_maps)} (not multiples of 16)")
        if int(feature_maps[0]) not in (16, 32):          # the head kernel's two instances (the ResUNet engine has a GEMM-fed head for wider first levels)
            unsupported(f"feature_maps[0] = {feature_maps[0]} (the output head reads 16 or 32 features)")
        self.compute_dtype = compute_dtype
        self._engine: Optional[UNetEngine] = None

        pool = nn.MaxPool2d if ndim == 2 else nn.MaxPool3d
        self.pre_upsampling = None
        self.conv_in = None
        self.down_path = nn.ModuleList()
        self.mpooling_layers = nn.ModuleList()
        c = in_ch
        for i in range(depth):
            self.down_path.append(ConvBlock(ndim, c, feature_maps[i], k_size, act, 2))
            self.mpooling_layers.append(pool((2, 2) if ndim == 2 else (zd[i], 2, 2)))
            c = feature_maps[i]
        self.bottleneck = ConvBlock(ndim, c, feature_maps[-1], k_size, act, 2)
        self.num_decoders = 1
        self.up_paths = nn.ModuleList([nn.ModuleList()])
        c = feature_maps[-1]
        for i in range(depth - 1, -1, -1):
            self.up_paths[0].append(UpBlock(ndim, c, feature_maps[i], feature_maps[i], k_size, act, zd[i], 2))
            c = feature_maps[i]
        self.conv_out = None
        self.post_upsampling = None
        self.heads = nn.Sequential()
        conv = nn.Conv2d if ndim == 2 else nn.Conv3d
        for oc in output_channels:
            self.heads.append(conv(feature_maps[0], oc, kernel_size=1, padding="same"))
        # blocks.py:2301-2336: Xavier-uniform + zero bias on Conv2d / Conv3d only (transposed convs keep PyTorch's default)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def engine(self) -> UNetEngine:
        if self._engine is None or self._engine.dtype != self.compute_dtype:
            self._engine = UNetEngine(self.cfg, self.ndim, self.compute_dtype)
        return self._engine

    def forward(self, x) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("biapy_amd.U_Net runs on the MI355X only (input is on %s); there is no CPU path" % x.device)
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        x = x.to(torch.float32)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _ResUNetFn.apply(x, self.engine(), names, *params)
        P = {n: p.detach() for n, p in zip(names, params)}
        logits, _ = self.engine().forward(P, x, head_act=0, save=False)
        return logits

    head_activation_code = ResUNet.head_activation_code
    _HEAD_CODES = ResUNet._HEAD_CODES

    @torch.no_grad()
    def predict_proba(self, x, head_activations=None) -> torch.Tensor:
        """Inference with the head activations (``ce_sigmoid`` by default; base_workflow.py:1403-1457) fused into the head kernel."""
        P = {n: p.detach() for n, p in self.named_parameters()}
        out, _ = self.engine().forward(P, x.to(torch.float32), head_act=self.head_activation_code(head_activations), save=False)
        return out
