"""Drop-in for ``biapy.models.resunet.ResUNet`` (3D) running on the MI355X engine.

Contract kept from the reference (biapy/models/resunet.py:34-60, :352-446; registry call at
biapy/models/__init__.py:120-147):
  * same constructor keyword arguments;
  * ``forward(x)`` takes ``(B,C,Z,Y,X)`` float32 (channels_last_3d strides, as ``to_pytorch_format``
    produces them) and returns the prediction tensor (logits when ``explicit_activations`` is False);
  * identical ``state_dict()`` keys and shapes so checkpoints load with
    ``strict=True`` both ways (biapy/utils/misc.py:611-630), and parameters are ordinary
    ``nn.Parameter``s so DistributedDataParallel / torchinfo / optimisers work unchanged.

The module tree below exists only to own the parameters under the reference's names; no nn.Module
forward of a leaf layer is ever called - ``forward`` hands the parameter tensors to
:class:`biapy_amd.engine.ResUNetEngine`, and gradients come from its hand-written backward through one
``torch.autograd.Function``.

2D networks and anisotropic levels (``MODEL.ISOTROPY[i] = False``: (1,3,3) kernels) run through the same 3x3x3 kernels with
zero-padded taps (``engine.lift_params``): correct, but 2/3 of those layers' MFMA work is spent on zeros - they are accepted
so that the drop-in covers the reference's templates, the tuned path is the isotropic 3D one.

Round 4: every per-element block activation of ``get_activation`` (blocks.py:1973-1998: relu, tanh, leaky_relu, elu, gelu, silu, sigmoid,
softplus, linear / none), the classification head (``"class"`` in ``output_channel_info``: ``forward`` returns ``{"pred", "class"}``) and
``explicit_activations`` (resunet.py:408-443) are built and pinned to fixtures generated from the reference class.

Configurations outside the accelerated hot path (normalisation other than "in" / "gn", larger_io, separated decoders, contrastive head,
``upsample_layer="upsampling"``, YX_DOWN != 2, Z_DOWN outside {1,2}, nconvs != 2, pre-activation order, softmax as a BLOCK
activation) raise ``NotImplementedError`` at construction: they stay on the reference's plain-PyTorch classes, selected by the same registry.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .engine import NetConfig, ResUNetEngine


def _act_layer(name: str) -> nn.Module:
    # the parameter-free placeholder modules of blocks.py:1986-1998 (they only keep the reference's module tree / state_dict numbering; the
    # activation itself runs in the kernels' prologues)
    layers = {"elu": lambda: nn.ELU(alpha=1.0, inplace=True), "relu": lambda: nn.ReLU(inplace=True), "silu": lambda: nn.SiLU(inplace=True),
              "leaky_relu": lambda: nn.LeakyReLU(inplace=True), "gelu": nn.GELU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "softplus": nn.Softplus,
              "linear": nn.Identity, "none": nn.Identity}
    if name not in layers:
        raise NotImplementedError(f"activation={name!r} is not implemented on the MI355X engine (softmax as a block activation is a channel reduction)")
    return layers[name]()


def _conv(ndim: int):
    return nn.Conv2d if ndim == 2 else nn.Conv3d


def _inorm(ndim: int, c: int, kind: str = "in") -> nn.Module:
    if kind == "gn":          # what blocks.py:2122-2125 means by 'gn' (its own call raises): GroupNorm(8, C); same parameter names and shapes
        return nn.GroupNorm(8, c)
    return (nn.InstanceNorm2d if ndim == 2 else nn.InstanceNorm3d)(c, affine=True, momentum=0.1)


class ConvBlock(nn.Module):
    """Parameter holder named like blocks.py:25-192: ``block = Sequential(conv[, norm, act])``.  ``k`` is an int or the
    reference's kernel tuple ((3,3) in 2D, (1,3,3) for an anisotropic level)."""

    def __init__(self, cin: int, cout: int, k, with_norm_act: bool, act: str, ndim: int = 3, norm: str = "in"):
        super().__init__()
        layers: List[nn.Module] = [_conv(ndim)(cin, cout, kernel_size=k, padding="same")]
        if with_norm_act:
            layers += [_inorm(ndim, cout, norm), _act_layer(act)]
        self.block = nn.Sequential(*layers)


class ResConvBlock(nn.Module):
    """Parameter holder named like blocks.py:1194-1459 (post-activation order, two convolutions)."""

    def __init__(self, cin: int, cout: int, k, act: str, first_block: bool, ndim: int = 3, norm: str = "in"):
        super().__init__()
        layers: List[nn.Module] = []
        if not first_block:
            layers += [_inorm(ndim, cin, norm), _act_layer(act)]
        layers += [ConvBlock(cin, cout, k, True, act, ndim, norm), ConvBlock(cout, cout, k, False, act, ndim, norm)]
        self.block = nn.Sequential(*layers)
        self.shortcut = nn.Sequential(_conv(ndim)(cin, cout, kernel_size=1, padding="same"))


class ResUpBlock(nn.Module):
    """Parameter holder named like blocks.py:1462-1655."""

    def __init__(self, cin: int, cbridge: int, cout: int, k, act: str, z_down: int = 2, ndim: int = 3, norm: str = "in"):
        super().__init__()
        if ndim == 2:
            self.up = nn.ConvTranspose2d(cin, cin, kernel_size=(2, 2), stride=(2, 2))
        else:
            self.up = nn.ConvTranspose3d(cin, cin, kernel_size=(z_down, 2, 2), stride=(z_down, 2, 2))   # k = s = (z_down, yx, yx), blocks.py:1607
        self.conv_block = ResConvBlock(cin + cbridge, cout, k, act, False, ndim, norm)


class _ResUNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine: ResUNetEngine, names: List[str], *params):
        P = dict(zip(names, (p.detach() for p in params)))
        need = any(p.requires_grad for p in params) and torch.is_grad_enabled()
        logits, saved = engine.forward(P, x.detach(), head_act=0, save=True)
        ctx.engine, ctx.names, ctx.saved = engine, names, saved
        ctx.params = P
        del need
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        G = ctx.engine.backward(ctx.params, ctx.saved, dlogits)
        ctx.saved = None
        return (None, None, None) + tuple(G[n] for n in ctx.names)


_SR_W1, _SR_WSC = "down_path.0.block.0.block.0.weight", "down_path.0.shortcut.0.weight"


def _sr_pre_forward(engine: ResUNetEngine, P, x, factor, head_act, save):
    """Super-resolution "pre" up-sampling (resunet.py:206-213, :368-369) + the network: the 1-channel image goes through
    ``bpx_upsample_c1_fwd`` into channel 0 of a 16-channel tensor and the first block's 1-input-channel weights are zero-padded to
    16 input channels, so that the generic kernels run the first block and hand back the gradient of the up-sampled image."""
    from . import _lib as L

    B, _, D, H, W = x.shape
    fz, fy, fx = factor
    img = x.reshape(B, D, H, W).contiguous()
    x16 = torch.empty((B, D * fz, H * fy, W * fx, 16), dtype=engine.dtype, device=x.device)
    L.check(L.lib.bpx_upsample_c1_fwd(engine.dt, B, D, H, W, fz, fy, fx, img.data_ptr(), P["pre_upsampling.weight"].contiguous().data_ptr(),
                                      P["pre_upsampling.bias"].data_ptr(), x16.data_ptr(), L.stream_ptr()))
    Pc = {k: v for k, v in P.items() if not k.startswith("pre_upsampling.")}
    for k in (_SR_W1, _SR_WSC):
        w = P[k]
        wp = torch.zeros((w.shape[0], 16) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device)
        wp[:, :1] = w
        Pc[k] = wp
    logits, saved = engine.forward(Pc, None, head_act=head_act, save=save, cache_weights=False, x_ndhwc=x16, want_dx=save)
    return logits, (saved, Pc, img)


class _ResUNetSRPreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine: ResUNetEngine, names: List[str], factor, *params):
        P = dict(zip(names, (p.detach() for p in params)))
        logits, (saved, Pc, img) = _sr_pre_forward(engine, P, x.detach(), factor, 0, True)
        ctx.engine, ctx.names, ctx.saved, ctx.Pc, ctx.img, ctx.factor, ctx.P = engine, names, saved, Pc, img, factor, P
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        from . import _lib as L

        eng = ctx.engine
        G = eng.backward(ctx.Pc, ctx.saved, dlogits)
        dx16 = G.pop("__dx__")
        fz, fy, fx = ctx.factor
        B, D, H, W = ctx.img.shape
        nb = L.lib.bpx_upsample_c1_blocks(B * D * H * W)
        part = torch.empty((fz * fy * fx, nb, 2), dtype=torch.float32, device=dx16.device)
        L.check(L.lib.bpx_upsample_c1_bwd(eng.gdt, B, D, H, W, fz, fy, fx, ctx.img.data_ptr(), dx16.data_ptr(), part.data_ptr(), L.stream_ptr()))
        sums = part.to(torch.float64).sum(1)
        out = {}
        for n in ctx.names:
            if n == "pre_upsampling.weight":
                out[n] = sums[:, 0].to(torch.float32).reshape(ctx.P[n].shape)
            elif n == "pre_upsampling.bias":
                out[n] = sums[:, 1].sum().to(torch.float32).reshape(1)
            elif n in (_SR_W1, _SR_WSC):
                out[n] = G[n][:, :1].contiguous()
            else:
                out[n] = G[n]
        ctx.saved = None
        return (None, None, None, None) + tuple(out[n] for n in ctx.names)


class _GraphedResUNetFn(torch.autograd.Function):
    """Forward / backward of the whole network as two HIP-graph replays (ResUNet.capture_graphs).  Anything around it -
    loss, DistributedDataParallel's gradient hooks and all-reduce, the optimizer - stays eager and unchanged."""

    @staticmethod
    def forward(ctx, x, gr, *params):
        gr["x"].copy_(x.detach(), non_blocking=True)
        gr["fwd"].replay()
        ctx.gr = gr
        return gr["logits"].detach()

    @staticmethod
    def backward(ctx, dlogits):
        gr = ctx.gr
        gr["dlogits"].copy_(dlogits, non_blocking=True)
        gr["bwd"].replay()
        return (None, None) + tuple(gr["grads"][n] for n in gr["names"])


class ResUNet(nn.Module):
    _bpx_dropin = True   # train_engine: the training-time model_call_func of this class is to_pytorch_format -> forward

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
        compute_dtype: torch.dtype = torch.float16,
    ):
        super().__init__()
        if len(output_channels) == 0:
            raise ValueError("'output_channels' needs to has at least one value")
        act = activation.lower()
        depth = len(feature_maps) - 1
        iso = [isotropy] * len(feature_maps) if isinstance(isotropy, bool) else list(isotropy)

        def unsupported(what):
            raise NotImplementedError(f"biapy_amd.ResUNet: {what} is outside the MI355X hot path; use the reference PyTorch class for it")

        ndim = 3 if len(image_shape) == 4 else 2
        if k_size != 3:
            unsupported("kernel size != 3")
        if list(yx_down)[:depth] != [2] * depth:
            unsupported("YX_DOWN other than 2")
        if ndim == 3 and (len(list(z_down)) < depth or any(int(v) not in (1, 2) for v in list(z_down)[:depth])):
            unsupported("Z_DOWN other than 1 or 2")
        if upsample_layer != "convtranspose":
            unsupported("upsample_layer != 'convtranspose'")
        if separated_decoders or contrast or larger_io:
            unsupported("separated decoders / contrastive head / larger_io")
        up = tuple(int(v) for v in upsampling_factor)
        self.sr_pre, self.sr_post = None, 0
        if len(up) > 0:
            # super-resolution (resunet.py:206-213 pre / :326-333 post): ConvTranspose with kernel = stride = the up-scaling factor
            if ndim != 3 or len(up) != 3 or not all(iso):
                unsupported("super-resolution up-sampling of 2D / anisotropic-kernel networks")
            if upsampling_position == "pre":
                if image_shape[-1] != 1 or any(v < 1 for v in up) or up[0] * up[1] * up[2] > 512:
                    unsupported("pre up-sampling of multi-channel images")
                self.sr_pre = up
            elif upsampling_position == "post":
                if up[1:] != (2, 2) or up[0] not in (1, 2):
                    unsupported(f"post up-sampling by {up} (the transposed-conv kernels are (1|2, 2, 2))")
                self.sr_post = up[0]
            else:
                raise ValueError(f"upsampling_position={upsampling_position!r}")
        if conv_block_order != "conv_norm_act" or list(conv_layers)[: depth + 1] != [2] * (depth + 1):
            unsupported("conv_block_order != 'conv_norm_act' or conv_layers != 2")
        dv = [float(d) for d in drop_values]
        if len(dv) < depth + 1 and any(d > 0 for d in dv):
            raise ValueError("'drop_values' needs one value per level and one for the bottleneck")
        self.depth = depth
        self.ndim = ndim
        self.z_down, self.yx_down = z_down, yx_down
        self.output_channels = output_channels
        self.output_channel_info = output_channel_info
        self.head_activations = list(head_activations)
        # Classification head (resunet.py:180, :408-443): an output head whose output_channel_info entry contains "class" leaves forward() as
        # out["class"], the others concatenated as out["pred"].  All heads are rows of ONE (sum(output_channels), fm0) matrix in the head
        # kernel; the split and - with explicit_activations - the per-channel activations of prepare_activation_layers (blocks.py:2001-2051) are
        # applied to the kernel's fp32 logits with differentiable PyTorch operators (one small elementwise pass per channel: the non-default
        # configuration), so the gradients of every head reach the head kernel's backward as one dlogits tensor.
        self.return_class = any("class" in str(info) for info in output_channel_info)
        self.contrast = False
        self.explicit_activations = bool(explicit_activations)
        self.return_one_tensor = return_one_tensor
        chan_info = [str(info) for info, n in zip(output_channel_info, output_channels) for _ in range(int(n))]
        if len(chan_info) != sum(output_channels):
            raise ValueError("'output_channel_info' needs one entry per output head")
        self._class_channels = [c for c, info in enumerate(chan_info) if "class" in info]
        self._pred_channels = [c for c, info in enumerate(chan_info) if "class" not in info]
        if not self._pred_channels:
            raise ValueError("at least one output head must not be a 'class' head")
        self._explicit_acts = None
        if self.explicit_activations:
            assert len(head_activations) == sum(output_channels), ("If 'explicit_activations' is True, 'head_activations' needs to have the same number "
                                                                   "of values as 'output_channels'")
            names_ = [a.lower().removeprefix("ce_") for a in head_activations]
            for a in names_:
                if a not in ("relu", "tanh", "leaky_relu", "elu", "gelu", "silu", "sigmoid", "softmax", "linear", "softplus", "none"):
                    raise AssertionError("Get unknown activation key {}".format(a))
            self._explicit_acts = names_
            # prepare_activation_layers (blocks.py:2001-2051): walk the channels in order, append each activation to the "pred" or the "class"
            # list, and STOP after the first entry whose name contains "softmax" - channels behind it get no activation at all
            self._pred_acts, self._class_acts = [], []
            for info, a in zip(chan_info, names_):
                (self._class_acts if "class" in info else self._pred_acts).append(a)
                if "softmax" in a:
                    break
            if self.return_class and not self._class_channels:
                raise ValueError("If 'return_class' is True, 'head_activations' must be provided.")
        # channel ranges of the class heads inside the class tensor (one activation per class HEAD, resunet.py:423-425)
        self._class_head_slices, o = [], 0
        for info, n in zip(output_channel_info, output_channels):
            if "class" in str(info):
                self._class_head_slices.append((o, o + int(n)))
                o += int(n)
        in_ch = image_shape[-1]
        zd = [int(v) for v in list(z_down)[:depth]] if ndim == 3 else [1] * depth
        self.cfg = NetConfig(in_ch=16 if self.sr_pre else in_ch, feature_maps=list(feature_maps), out_channels=tuple(output_channels), activation=act,
                             normalization=normalization, z_down=zd, ndim=ndim, post_up=self.sr_post,
                             dropout=(dv[:depth] + [dv[-1]]) if any(d > 0 for d in dv) else None)
        # kernel of level i (resunet.py:239-241, :260-262, :282-284): (3,3) in 2D, (1,3,3) where MODEL.ISOTROPY[i] is False
        ks = [(3, 3) if ndim == 2 else ((3, 3, 3) if iso[i] else (1, 3, 3)) for i in range(depth + 1)]
        self.compute_dtype = compute_dtype
        self._engine: Optional[ResUNetEngine] = None

        self.pre_upsampling = nn.ConvTranspose3d(in_ch, in_ch, kernel_size=self.sr_pre, stride=self.sr_pre) if self.sr_pre else None
        self.conv_in = None
        self.down_path = nn.ModuleList()
        self.mpooling_layers = nn.ModuleList()
        c = in_ch
        for i in range(depth):
            self.down_path.append(ResConvBlock(c, feature_maps[i], ks[i], act, first_block=(i == 0), ndim=ndim, norm=normalization))
            self.mpooling_layers.append(nn.MaxPool2d((2, 2)) if ndim == 2 else nn.MaxPool3d((zd[i], 2, 2)))
            c = feature_maps[i]
        self.bottleneck = ResConvBlock(c, feature_maps[-1], ks[-1], act, False, ndim, normalization)
        self.num_decoders = 1
        self.up_paths = nn.ModuleList([nn.ModuleList()])
        c = feature_maps[-1]
        for i in range(depth - 1, -1, -1):
            self.up_paths[0].append(ResUpBlock(c, feature_maps[i], feature_maps[i], ks[i], act, zd[i], ndim, normalization))
            c = feature_maps[i]
        self.conv_out = None
        self.post_upsampling = (nn.ConvTranspose3d(feature_maps[0], feature_maps[0], kernel_size=(self.sr_post, 2, 2), stride=(self.sr_post, 2, 2))
                                if self.sr_post else None)
        self.heads = nn.Sequential()
        for oc in output_channels:
            self.heads.append(_conv(ndim)(feature_maps[0], oc, kernel_size=1, padding="same"))
        self._init_weights()

    def _init_weights(self):
        # blocks.py:2301-2336: Xavier-uniform + zero bias on Conv3d only (ConvTranspose3d keeps PyTorch's default)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------------------------------
    def engine(self) -> ResUNetEngine:
        # one engine per storage type: switching compute_dtype for an evaluation pass and back keeps both engines' packed-weight caches
        if self._engine is None or self._engine.dtype != self.compute_dtype:
            if not hasattr(self, "_engines"):
                self._engines = {}
            if self.compute_dtype not in self._engines:
                self._engines[self.compute_dtype] = ResUNetEngine(self.cfg, self.compute_dtype)
            self._engine = self._engines[self.compute_dtype]
        self._engine.drop_active = bool(self.training)      # nn.Dropout semantics: masks in training mode only (blocks.py:163)
        return self._engine

    def train(self, mode: bool = True):
        """Entering training mode drops the packed / lifted weight copies the inference path keeps (they are also keyed by
        tensor version and by engine.weights_epoch(), which every graph replay of this package bumps)."""
        for eng in getattr(self, "_engines", {}).values():
            eng.clear_caches()
        return super().train(mode)

    def _named(self):
        names, params = [], []
        for n, p in self.named_parameters():
            names.append(n)
            params.append(p)
        return names, params

    def forward(self, x) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("biapy_amd.ResUNet runs on the MI355X only (input is on %s); there is no CPU path" % x.device)
        names, params = self._named()
        x = x.to(torch.float32)
        if self.sr_pre:
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                return _ResUNetSRPreFn.apply(x, self.engine(), names, self.sr_pre, *params)
            P = {n: p.detach() for n, p in zip(names, params)}
            logits, _ = _sr_pre_forward(self.engine(), P, x, self.sr_pre, 0, False)
            return logits
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            gr = getattr(self, "_graphs", None)
            if gr is not None and tuple(x.shape) == gr["shape"] and x.stride() == gr["stride"] and not torch.cuda.is_current_stream_capturing():
                return self._finish_outputs(_GraphedResUNetFn.apply(x, gr, *params))
            return self._finish_outputs(_ResUNetFn.apply(x, self.engine(), names, *params))
        P = {n: p.detach() for n, p in zip(names, params)}
        logits, _ = self.engine().forward(P, x, head_act=0, save=False, cache_weights=not self.training)
        return self._finish_outputs(logits)

    @staticmethod
    def _apply_named(t: torch.Tensor, name: str) -> torch.Tensor:
        import torch.nn.functional as F_
        fn = {"relu": torch.relu, "tanh": torch.tanh, "leaky_relu": F_.leaky_relu, "elu": F_.elu, "gelu": F_.gelu, "silu": F_.silu, "sigmoid": torch.sigmoid,
              "softmax": lambda v: torch.softmax(v, dim=1), "linear": lambda v: v, "softplus": F_.softplus, "none": lambda v: v}[name]
        return fn(t)

    def _finish_outputs(self, logits: torch.Tensor):
        """The tail of the reference forward (resunet.py:408-443) on the head kernel's (B, sum(output_channels), ...) logits."""
        if not self.return_class and not self.explicit_activations:
            return logits
        outs = logits[:, self._pred_channels] if self.return_class else logits
        cls = logits[:, self._class_channels] if self.return_class else None
        if self.explicit_activations:
            # resunet.py:415-425 with the lists of prepare_activation_layers: ONE collected activation acts on the whole tensor (a joint softmax over
            # the N channels of a multi-class head); several act on their own 1-channel slice each (a softmax there is over one channel), and the
            # channels behind the last collected entry stay raw logits
            pa = self._pred_acts
            if len(pa) == 1:
                outs = self._apply_named(outs, pa[0])
            elif len(pa) > 1:
                outs = torch.cat([self._apply_named(outs[:, i:i + 1], a) for i, a in enumerate(pa)] + ([outs[:, len(pa):]] if len(pa) < outs.shape[1] else []), dim=1)
            if cls is not None and self._class_acts:
                # entry i acts on the whole output of class head i (class_outs[i]); more entries than class heads is the reference's IndexError
                if len(self._class_acts) > len(self._class_head_slices):
                    raise IndexError("list index out of range")
                parts = [cls[:, a:b] for a, b in self._class_head_slices]
                for i, a in enumerate(self._class_acts):
                    parts[i] = self._apply_named(parts[i], a)
                cls = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]
        if not self.return_class:
            return outs
        if self.return_one_tensor:
            return torch.cat((outs, torch.argmax(cls, dim=1).unsqueeze(1)), dim=1)
        return {"pred": outs, "class": cls}

    def capture_graphs(self, x_example: torch.Tensor, warmup: int = 3) -> None:
        """Capture the training forward and backward for inputs shaped like ``x_example`` into two HIP graphs; later
        training-mode calls with that shape replay them (two launches instead of ~170: an eager step is host-bound, 16.3 vs
        13.8 ms measured).  Unlike ``graphs.GraphedTrainStep`` this leaves the autograd boundary intact, so it composes with
        ``DistributedDataParallel`` (wrap AFTER capturing): gradient hooks, bucketed all-reduce and the optimizer run as
        usual.  Parameters must keep their storage (in-place optimizer updates and ``load_state_dict`` do).  ``release_graphs()``
        drops the graphs and their private memory pool."""
        if not x_example.is_cuda:
            raise RuntimeError("capture_graphs needs a CUDA/HIP example input")
        names, params = self._named()
        eng = self.engine()
        P = {n: p.detach() for n, p in zip(names, params)}
        xs = x_example.detach().to(torch.float32).clone()

        def once():
            logits, saved = eng.forward(P, xs, head_act=0, save=True)
            eng.backward(P, saved, torch.ones_like(logits))

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                once()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: a HIP call from another thread (e.g. the RCCL watchdog of an initialised process group) must not
        # invalidate the capture
        with torch.cuda.graph(gf, capture_error_mode="thread_local"):
            logits, saved = eng.forward(P, xs, head_act=0, save=True)
        dl = torch.zeros_like(logits)
        with torch.cuda.graph(gb, pool=gf.pool(), capture_error_mode="thread_local"):
            grads = eng.backward(P, saved, dl)
        torch.cuda.synchronize()
        self._graphs = dict(shape=tuple(xs.shape), stride=xs.stride(), x=xs, logits=logits, saved=saved, dlogits=dl, grads=grads,
                            names=names, fwd=gf, bwd=gb)

    def release_graphs(self) -> None:
        self._graphs = None

    supported_compute_dtypes = (torch.float32, torch.bfloat16, torch.float16)   # float16: fp16 forward, bf16 gradients (engine.py)

    _HEAD_CODES = {"linear": 0, "ce_sigmoid": 1, "sigmoid": 1, "tanh": 2, "ce_softmax": 3, "softmax": 3}

    def head_activation_code(self, head_activations=None) -> int:
        """One 4-bit code per output channel for the head kernel (0 linear, 1 sigmoid, 2 tanh, 3 softmax; consecutive softmax
        channels are one group) from the reference's per-channel activation names (``apply_model_activations``,
        base_workflow.py:1403-1457).  A list shorter than the channel count repeats its last entry (one name per head)."""
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
        """Inference with the head activations (``ce_sigmoid`` by default; base_workflow.py:1403-1457) fused into the head kernel."""
        names, params = self._named()
        P = {n: p.detach() for n, p in zip(names, params)}
        if self.sr_pre:
            return _sr_pre_forward(self.engine(), P, x.to(torch.float32), self.sr_pre, self.head_activation_code(head_activations), False)[0]
        out, _ = self.engine().forward(P, x.to(torch.float32), head_act=self.head_activation_code(head_activations), save=False,
                                       cache_weights=True)
        return out
