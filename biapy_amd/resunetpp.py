"""Drop-in for ``biapy.models.resunet++.ResUNetPlusPlus`` (3D) running on the MI355X engine (cfg 4).

Contract kept from the reference (biapy/models/resunet++.py:40-513; registry call at biapy/models/__init__.py:120-149 with
``modelname = "resunet++"`` -> class ``ResUNetPlusPlus``): same constructor keyword arguments, ``forward(x)`` takes ``(B,C,Z,Y,X)``
float32 and returns the prediction tensor (the head activations - ``ce_sigmoid`` for the B / C channels, ``tanh`` for D - are
applied by the workflow, base_workflow.py:1403-1457, or fused into ``biapy_amd.losses.InstanceChannelsLoss``), identical
``state_dict()`` keys and shapes (tests/golden/resunetpp_golden.npz holds the reference's), ordinary ``nn.Parameter``s so
DistributedDataParallel / optimisers / checkpoints work unchanged.

The module tree below exists only to own the parameters under the reference's names; ``forward`` hands them to
:class:`biapy_amd.resunetpp_engine.ResUNetPPEngine`, gradients come from its hand-written backward through one autograd.Function.
Configurations outside the accelerated path (2D, normalisation other than "in", larger_io, separated decoders, contrastive head,
SR up-sampling, YX_DOWN != 2, nconvs != 2, pre-activation order, dropout, anisotropic kernels) raise ``NotImplementedError`` at
construction and stay on the reference class, selected by the same registry.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .resunetpp_engine import PPConfig, ResUNetPPEngine


def _act(name: str) -> nn.Module:
    return {"elu": nn.ELU(alpha=1.0, inplace=True), "relu": nn.ReLU(inplace=True), "silu": nn.SiLU(inplace=True)}[name]


def _in(c: int) -> nn.Module:
    return nn.InstanceNorm3d(c, affine=True, momentum=0.1)


class _ConvBlock(nn.Module):
    def __init__(self, cin, cout, with_norm_act, act):
        super().__init__()
        layers: List[nn.Module] = [nn.Conv3d(cin, cout, kernel_size=3, padding="same")]
        if with_norm_act:
            layers += [_in(cout), _act(act)]
        self.block = nn.Sequential(*layers)


class _ResConvBlock(nn.Module):
    """blocks.py:1194-1459 with ``skip_k_size = 3`` and ``skip_norm``: shortcut = Sequential(conv3, norm)."""

    def __init__(self, cin, cout, act, first_block):
        super().__init__()
        layers: List[nn.Module] = []
        if not first_block:
            layers += [_in(cin), _act(act)]
        layers += [_ConvBlock(cin, cout, True, act), _ConvBlock(cout, cout, False, act)]
        self.block = nn.Sequential(*layers)
        self.shortcut = nn.Sequential(nn.Conv3d(cin, cout, kernel_size=3, padding="same"), _in(cout))


class _SqEx(nn.Module):
    def __init__(self, c, r=16):
        super().__init__()
        self.excitation = nn.Sequential(nn.Linear(c, c // r, bias=False), nn.ReLU(inplace=True), nn.Linear(c // r, c, bias=False), nn.Sigmoid())


class _ASPP(nn.Module):
    def __init__(self, cin, cout, rates=(6, 12, 18)):
        super().__init__()
        for j, d in enumerate(rates):
            setattr(self, f"aspp_block{j + 1}", nn.Sequential(nn.Conv3d(cin, cout, 3, stride=1, padding=d, dilation=d), nn.ReLU(inplace=True), _in(cout)))
        self.output = nn.Conv3d(len(rates) * cout, cout, 1)


class _Attention(nn.Module):
    def __init__(self, c_enc, c_dec, c_out, z_down):
        super().__init__()
        self.conv_encoder = nn.Sequential(_in(c_enc), nn.ReLU(), nn.Conv3d(c_enc, c_out, 3, padding=1), nn.MaxPool3d((z_down, 2, 2)))
        self.conv_decoder = nn.Sequential(_in(c_dec), nn.ReLU(), nn.Conv3d(c_dec, c_out, 3, padding=1))
        self.conv_attn = nn.Sequential(_in(c_out), nn.ReLU(), nn.Conv3d(c_out, 1, 1))


class _ResUpBlock(nn.Module):
    def __init__(self, cin, cbridge, cout, act, z_down):
        super().__init__()
        self.up = nn.ConvTranspose3d(cin, cin, kernel_size=(z_down, 2, 2), stride=(z_down, 2, 2))
        self.conv_block = _ResConvBlock(cin + cbridge, cout, act, False)


class _PPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine: ResUNetPPEngine, names: List[str], *params):
        P = dict(zip(names, (p.detach() for p in params)))
        logits, saved = engine.forward(P, x.detach(), head_act=0, save=True)
        ctx.engine, ctx.names, ctx.saved, ctx.params = engine, names, saved, P
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        G = ctx.engine.backward(ctx.params, ctx.saved, dlogits)
        ctx.saved = None
        return (None, None, None) + tuple(G[n] for n in ctx.names)


class ResUNetPlusPlus(nn.Module):
    _bpx_dropin = True   # train_engine: the training-time model_call_func of this class is to_pytorch_format -> forward (+ head activations)
    # float16 (round 4) = the mixed training mode: fp16 forward / activations, bf16 gradients (resunetpp_engine.py on the gdt / bdt codes of engine.py)
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
        z_down=[2, 2, 2, 2],
        yx_down=[2, 2, 2, 2],
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
        conv_block_order: str = "conv_norm_act",
        contrast: bool = False,
        contrast_proj_dim: int = 256,
        return_one_tensor: bool = False,
        compute_dtype: torch.dtype = torch.bfloat16,
    ):
        super().__init__()
        if len(output_channels) == 0:
            raise ValueError("'output_channels' needs to has at least one value")
        act = activation.lower()
        depth = len(feature_maps) - 2
        iso = [isotropy] * len(feature_maps) if isinstance(isotropy, bool) else list(isotropy)

        def unsupported(what):
            raise NotImplementedError(f"biapy_amd.ResUNetPlusPlus: {what} is outside the MI355X hot path; use the reference PyTorch class for it")

        if len(image_shape) != 4:
            unsupported("the 2D network")
        if normalization != "in":
            unsupported(f"normalization={normalization!r}")
        if k_size != 3 or not all(iso[: depth + 1]):
            unsupported("kernel sizes other than (3,3,3)")
        if list(yx_down)[1: depth + 1] != [2] * depth:
            unsupported("YX_DOWN other than 2")
        if upsample_layer != "convtranspose":
            unsupported("upsample_layer != 'convtranspose'")
        if separated_decoders or contrast or larger_io or len(upsampling_factor) > 0:
            unsupported("separated decoders / contrastive head / larger_io / super-resolution up-sampling")
        if conv_block_order != "conv_norm_act" or list(conv_layers)[: depth + 1] != [2] * (depth + 1):
            unsupported("conv_block_order != 'conv_norm_act' or conv_layers != 2")
        if any(float(d) > 0 for d in list(drop_values)[: depth + 2]):
            unsupported("dropout")
        if explicit_activations or "class" in output_channel_info:
            unsupported("explicit head activations / classification head")
        self.depth, self.ndim = depth, 3
        self.z_down, self.yx_down = z_down, yx_down
        self.output_channels, self.output_channel_info = output_channels, output_channel_info
        self.head_activations = list(head_activations)
        self.return_class, self.contrast, self.explicit_activations, self.return_one_tensor = False, False, False, return_one_tensor
        fm = list(feature_maps)
        zd = [int(v) for v in list(z_down)[: depth + 1]]
        self.cfg = PPConfig(in_ch=image_shape[-1], feature_maps=fm, out_channels=tuple(output_channels), activation=act, z_down=zd)
        self.compute_dtype = compute_dtype
        self._engine: Optional[ResUNetPPEngine] = None

        self.pre_upsampling = None
        self.conv_in = None
        self.down_path = nn.ModuleList([_ResConvBlock(image_shape[-1], fm[0], act, True)])
        self.mpooling_layers = nn.ModuleList([nn.MaxPool3d((zd[0], 2, 2))])
        self.sqex_blocks = nn.ModuleList([_SqEx(fm[0])])
        c = fm[0]
        for i in range(depth):
            self.down_path.append(_ResConvBlock(c, fm[i + 1], act, False))
            self.mpooling_layers.append(nn.MaxPool3d((zd[i + 1], 2, 2)))
            c = fm[i + 1]
            if i != depth - 1:
                self.sqex_blocks.append(_SqEx(c))
        self.sqex_blocks.append(None)                      # resunet++.py:303-305: so that zip() over the three lists works
        self.aspp_bridge = _ASPP(c, fm[-1])
        self.num_decoders = 1
        self.up_paths = nn.ModuleList([nn.ModuleList()])
        self.attentions = nn.ModuleList([nn.ModuleList()])
        c = fm[-1]
        for i in range(depth - 1, -1, -1):
            self.attentions[0].append(_Attention(fm[i], c, c, zd[i + 1]))
            self.up_paths[0].append(_ResUpBlock(c, fm[i], fm[i + 1], act, zd[i + 1]))
            c = fm[i + 1]
        self.aspp_out = nn.ModuleList([_ASPP(fm[1], fm[0])])
        self.conv_out = None
        self.post_upsampling = None
        self.heads = nn.Sequential()
        for oc in output_channels:
            self.heads.append(nn.Conv3d(fm[0], oc, kernel_size=1, padding="same"))
        # blocks.py:2301-2336: Xavier-uniform + zero bias on Conv3d only (ConvTranspose3d / Linear keep PyTorch's default)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def engine(self) -> ResUNetPPEngine:
        if self._engine is None or self._engine.dtype != self.compute_dtype:
            self._engine = ResUNetPPEngine(self.cfg, self.compute_dtype)
        return self._engine

    def _named(self):
        names, params = [], []
        for n, p in self.named_parameters():
            names.append(n)
            params.append(p)
        return names, params

    def forward(self, x) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("biapy_amd.ResUNetPlusPlus runs on the MI355X only (input is on %s); there is no CPU path" % x.device)
        names, params = self._named()
        x = x.to(torch.float32)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _PPFn.apply(x, self.engine(), names, *params)
        P = {n: p.detach() for n, p in zip(names, params)}
        logits, _ = self.engine().forward(P, x, head_act=0, save=False)
        return logits

    _HEAD_CODES = {"linear": 0, "ce_sigmoid": 1, "sigmoid": 1, "tanh": 2, "ce_softmax": 3, "softmax": 3}

    def head_activation_code(self, head_activations=None) -> int:
        acts = [a.lower() for a in (head_activations if head_activations is not None else self.head_activations)]
        n_out = sum(self.output_channels)
        acts = (acts + [acts[-1]] * n_out)[:n_out]
        code = 0
        for c, a in enumerate(acts):
            if a not in self._HEAD_CODES:
                raise NotImplementedError(f"head activation {a!r} is not implemented in the MI355X head kernel")
            code |= self._HEAD_CODES[a] << (4 * c)
        return code

    @torch.no_grad()
    def predict_proba(self, x, head_activations=None) -> torch.Tensor:
        """Inference with the head activations (base_workflow.py:1403-1457) fused into the head kernel."""
        names, params = self._named()
        P = {n: p.detach() for n, p in zip(names, params)}
        out, _ = self.engine().forward(P, x.to(torch.float32), head_act=self.head_activation_code(head_activations), save=False)
        return out
