"""ctypes binding of libbiapy_amd.so (include/biapy_amd.h).

The HIP extension IS the product: there is no CPU or PyTorch fallback.  Importing this module
without the built library raises immediately (``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C biapy_amd/csrc`` builds it in-tree).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BPX_LIB_PATH: A/B aid - another build of the same library (scripts/ab_build.sh builds the kernel sources of a git revision next to the tree's)
LIB_PATH = os.environ.get("BPX_LIB_PATH") or os.path.join(_HERE, "libbiapy_amd.so")

F32, BF16, F16, U8, MIX16 = 0, 1, 2, 3, 4   # MIX16: backward entries only - fp16 activations, bf16 gradients (include/biapy_amd.h)
# block activations of the reference (blocks.py:1973-1998): every entry of get_activation except "softmax" (a channel reduction, not a per-element prologue)
ACT = {"none": 0, "linear": 0, "elu": 1, "relu": 2, "silu": 3, "leaky_relu": 4, "gelu": 5, "tanh": 6, "sigmoid": 7, "softplus": 8}
PK_K3, PK_K3_T, PK_K1, PK_DENSE, PK_DENSE_T, PK_CT, PK_CT_T, PK_CT4, PK_CT4_T = range(9)


class AxisGrid(C.Structure):
    _fields_ = [("n", C.c_int32), ("step", C.c_int32), ("last", C.c_int32), ("patch", C.c_int32), ("limit", C.c_int32)]


class Tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int32), ("C", C.c_int32), ("cs", C.c_int64)]


NULL_T = Tensor(None, 0, 0, 0)

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

_SIGS = {
    "bpx_version": ([], C.c_int),
    "bpx_last_error": ([], C.c_char_p),
    "bpx_selftest_layouts": ([_vp, _vp], _i),
    "bpx_debug_set_wgrad_tr": ([_i], _i),
    "bpx_debug_set_conv_ws": ([_i], _i),
    "bpx_debug_set_conv_stamps": ([_vp], _i),
    "bpx_debug_set_tiling_scalar": ([_i], _i),
    "bpx_debug_set_conv_occ": ([_i], _i),
    "bpx_debug_set_conv_zm": ([_i], _i),
    "bpx_debug_conv_zm_launches": ([], _i),
    "bpx_debug_conv_zm_occupancy": ([_i], _i),
    "bpx_crop3d_gather": ([_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(AxisGrid), _i64, _i64, _vp, _vp], _i),
    "bpx_merge3d_blend": ([_vp, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(AxisGrid), _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i,
                           _vp, _vp, _i, _vp, _i, _vp], _i),
    "bpx_packed_weight_elems": ([_i, _i, _i, _i], _i64),
    "bpx_pack_weight": ([_i, _vp, _i, _i, _i, _vp, _vp], _i),
    "bpx_pack_weights_batched": ([_i, _i, _vp, _vp], _i),
    "bpx_adam_step": ([_i, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i, _vp], _i),
    "bpx_scan_blocks": ([_i64], _i),
    "bpx_select_workspace": ([], _i64),
    "bpx_select_kth_f32": ([_vp, _i64, _i64, _vp, _vp, _vp], _i),
    "bpx_minmax_f32": ([_vp, _i64, _vp, _vp], _i),
    "bpx_moment_f32": ([_vp, _i64, C.c_double, _i, _vp, _vp], _i),
    "bpx_histogram_f32": ([_vp, _i64, _f, _f, _i, _vp, _vp, _vp], _i),
    "bpx_threshold_u8": ([_vp, _i64, _f, _vp, _vp], _i),
    "bpx_clip_affine_f32": ([_vp, _i64, _f, _f, _f, _f, _vp, _vp], _i),
    "bpx_class_argmax": ([_vp, _i64, _i, _i, _vp, _vp], _i),
    "bpx_tta_orient": ([_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp], _i),
    "bpx_tta_accumulate": ([_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp], _i),
    "bpx_chan_loss_blocks": ([_i64], _i),
    "bpx_chan_loss_sums": ([_vp, _vp, _i, _i, _i64, C.c_uint, _vp, _vp], _i),
    "bpx_chan_loss_bwd": ([_vp, _vp, _i, _i, _i64, C.c_uint, _vp, _vp, _vp], _i),
    "bpx_gate_mul_fwd": ([_i, _i64, Tensor, Tensor, Tensor, _vp], _i),
    "bpx_gate_mul_bwd": ([_i, _i64, Tensor, Tensor, Tensor, Tensor, _vp, _vp], _i),
    "bpx_seg_loss_blocks": ([_i64], _i),
    "bpx_seg_loss_sums": ([_vp, _vp, _i64, _vp, _vp], _i),
    "bpx_seg_loss_bwd": ([_vp, _vp, _i64, _vp, _vp, _vp], _i),
    "bpx_seg_loss_finish": ([_vp, _i, _i64, _f, _f, _f, _vp, _vp, _vp], _i),
    "bpx_seg_loss_bwd_fused": ([_vp, _vp, _i64, _vp, _vp, _f, _f, _f, _vp, _vp], _i),
    "bpx_softmax_ce_blocks": ([_i64], _i),
    "bpx_softmax_ce_row": ([], _i),
    "bpx_softmax_ce_sums": ([_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp], _i),
    "bpx_softmax_ce_finish": ([_vp, _i, _i64, _vp, _vp, _vp], _i),
    "bpx_softmax_ce_bwd": ([_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "bpx_chan_loss_finish": ([_vp, _i, _i, _i64, _vp, _vp, _vp], _i),
    "bpx_chan_loss_bwd_fused": ([_vp, _vp, _i, _i, _i64, C.c_uint, _vp, _vp, _vp, _vp], _i),
    "bpx_conv3d_fwd": ([_i, _i, _i, _i, _i, Tensor, _vp, _i, _vp, _vp, Tensor, _vp, _vp, Tensor, _vp, _vp], _i),
    "bpx_conv3d_fwd_pool": ([_i, _i, _i, _i, _i, Tensor, _vp, _i, _vp, _vp, Tensor, _vp, _vp, Tensor, _vp, _i, Tensor, _vp, _vp], _i),
    "bpx_conv3d_fwd_pool_supported": ([_i, _i, _i, _i, _i, _i, _i, _i], _i),
    "bpx_conv3d_fwd_shuffle": ([_i, _i, _i, _i, _i, Tensor, _vp, _i, _vp, _vp, _i, Tensor, _vp], _i),
    "bpx_conv3d_stats_tiles": ([_i, _i, _i, _i, _i, _i], _i),
    "bpx_conv3d_dgrad": ([_i, _i, _i, _i, _i, Tensor, _vp, Tensor, _vp, _i, Tensor, _vp, _vp], _i),
    "bpx_conv3d_wgrad": ([_i, _i, _i, _i, _i, Tensor, _vp, _i, Tensor, _i, _vp, _vp, _vp, _i64, _vp], _i),
    "bpx_conv3d_wgrad_db2": ([_i, _i, _i, _i, _i, Tensor, _vp, _i, Tensor, _i, _vp, _vp, _vp, _vp, _i64, _vp], _i),
    "bpx_conv3d_wgrad_workspace": ([_i, _i, _i, _i, _i, _i, _i], _i64),
    "bpx_conv3d_bwd_fused_supported": ([_i, _i, _i, _i, _i, _i, _i], _i),
    "bpx_conv3d_bwd_fused_stats_tiles": ([_i, _i, _i, _i, _i, _i], _i),
    "bpx_conv3d_bwd_fused_workspace": ([_i, _i, _i, _i, _i, _i], _i64),
    "bpx_conv3d_bwd_fused": ([_i, _i, _i, _i, _i, Tensor, _vp, Tensor, _vp, _i, Tensor, _vp, _vp, _vp, _vp, _vp, _i64, _vp], _i),
    "bpx_debug_set_bwd_fused": ([_i], _i),
    "bpx_debug_set_bwd_rs": ([_i], _i),
    "bpx_debug_set_conv_kg": ([_i], _i),
    "bpx_debug_set_tile_order": ([_i], _i),
    "bpx_debug_set_wgrad_cap": ([_i], _i),
    "bpx_debug_set_wgrad_k1": ([_i], _i),
    "bpx_debug_set_pw_stream": ([_i], _i),
    "bpx_debug_set_convt_k1": ([_i], _i),
    "bpx_debug_set_c1_persist": ([_i], _i),
    "bpx_convT3d_k2s2_wgrad_workspace": ([_i, _i, _i, _i, _i, _i, _i], _i64),
    "bpx_conv1x1_fwd": ([_i, _i, _i64, Tensor, _vp, _vp, Tensor, Tensor, _vp, Tensor, Tensor, _vp], _i),
    "bpx_conv1x1_fwd_split_wgrad_workspace": ([_i, _i, _i64, _i], _i64),
    "bpx_conv1x1_fwd_split_wgrad": ([_i, _i, _i64, Tensor, _vp, Tensor, Tensor, _vp, Tensor, Tensor, _vp, _vp, _i64, _vp], _i),
    "bpx_conv1x1_fwd_split": ([_i, _i, _i64, Tensor, _vp, _vp, Tensor, Tensor, _vp, Tensor, Tensor, Tensor, _vp], _i),
    "bpx_convT3d_k2s2_fwd": ([_i, _i, _i, _i, _i, _i, Tensor, _vp, _vp, Tensor, _vp, _vp], _i),
    "bpx_convT3d_stats_tiles": ([_i, _i, _i, _i], _i),
    "bpx_convT3d_k2s2_dgrad": ([_i, _i, _i, _i, _i, _i, Tensor, _vp, Tensor, _vp], _i),
    "bpx_convT3d_k2s2_wgrad": ([_i, _i, _i, _i, _i, _i, Tensor, Tensor, _vp, _vp, _vp, _i64, _vp], _i),
    "bpx_norm_finalize": ([_vp, _i, _i, _i, _i64, _vp, _vp, _f, _i, _vp, _i, _i, _vp], _i),
    "bpx_norm_channel_sums": ([_vp, _i, _i, _i, _vp, _i, _i, _vp], _i),
    "bpx_groupnorm_finalize": ([_vp, _i, _i, _i64, _vp, _vp, _f, _i, _vp, _vp], _i),
    "bpx_groupnorm_bwd_finalize": ([_vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp], _i),
    "bpx_tensor_stats": ([_i, _i, _i64, Tensor, _vp, _vp], _i),
    "bpx_tensor_stats_tiles": ([_i64], _i),
    "bpx_norm_bwd_finalize": ([_vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp], _i),
    "bpx_norm_bwd_finalize_deferred": ([_vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp], _i),
    "bpx_gather3d_tables": ([_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp], _i),
    "bpx_scatter3d_tables": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp], _i),
    "bpx_scatter3d_regions": ([_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp], _i),
    "bpx_wgrad_defer_begin": ([], _i),
    "bpx_wgrad_defer_flush": ([_vp], _i),
    "bpx_norm_bwd_apply": ([_i, _i, _i64, Tensor, Tensor, _vp, Tensor, Tensor, _vp], _i),
    "bpx_channel_affine": ([_i, _i, _i64, Tensor, Tensor, _vp, _vp, Tensor, _vp], _i),
    "bpx_dot_stats": ([_i, _i, _i64, Tensor, Tensor, _vp, _vp], _i),
    "bpx_gate_mlp_fwd": ([_vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp], _i),
    "bpx_gate_mlp_bwd": ([_vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "bpx_norm_act_tiles": ([_i, _i64, _i], _i),
    "bpx_norm_act_dropout_tiles": ([_i, _i64, _i], _i),
    "bpx_norm_act_dropout_fwd": ([_i, _i, _i64, Tensor, _vp, _i, C.c_float, C.c_uint64, _vp, _i, _vp, _i, Tensor, _vp], _i),
    "bpx_norm_act_dropout_bwd": ([_i, _i, _i64, Tensor, Tensor, _vp, _i, C.c_float, C.c_uint64, _vp, _i, _vp, _i, Tensor, _vp, _vp], _i),
    "bpx_norm_act_fwd": ([_i, _i, _i64, Tensor, _vp, _i, Tensor, _vp], _i),
    "bpx_norm_act_bwd": ([_i, _i, _i64, Tensor, Tensor, _vp, _i, Tensor, Tensor, _vp, _vp], _i),
    "bpx_maxpool3d_fwd": ([_i, _i, _i, _i, _i, _i, Tensor, Tensor, _vp, _vp], _i),
    "bpx_maxpool3d_stats_tiles": ([_i, _i, _i, _i, _i, _i], _i),
    "bpx_maxpool3d_bwd_r1_workspace": ([_i, _i, _i, _i, _i, _i, _i], _i64),
    "bpx_maxpool3d_bwd_r1": ([_i, _i, _i, _i, _i, _i, Tensor, Tensor, Tensor, Tensor, _vp, _vp, _vp, _i64, _vp], _i),
    "bpx_maxpool3d_bwd": ([_i, _i, _i, _i, _i, _i, Tensor, Tensor, Tensor, Tensor, _vp], _i),
    "bpx_head_fwd": ([_i, _i64, _i, Tensor, _vp, _vp, _i, _i, _vp, _i64, _i64, _vp], _i),
    "bpx_head_bwd": ([_i, _i64, _i, Tensor, _vp, _i, _vp, _i64, _i64, Tensor, _vp, _vp, _vp, _i64, _vp], _i),
    "bpx_head_bwd_workspace": ([_i, _i], _i64),
    "bpx_conv3d_c1_wgrad_workspace": ([_i], _i64),
    "bpx_conv1x1_c1_wgrad_workspace": ([_i], _i64),
    "bpx_conv3d_c1_fwd": ([_i, _i, _i, _i, _i, _vp, _vp, _vp, Tensor, _vp, _vp], _i),
    "bpx_conv3d_c1_stats_tiles": ([_i, _i, _i], _i),
    "bpx_conv3d_c1_wgrad": ([_i, _i, _i, _i, _i, _vp, Tensor, _vp, _vp, _vp, _i64, _vp], _i),
    "bpx_conv3d_c1_wgrad_nb_supported": ([_i, _i], _i),
    "bpx_conv3d_c1_wgrad_nb": ([_i, _i, _i, _i, _i, _vp, Tensor, Tensor, _vp, _vp, _vp, _vp, _i64, _vp], _i),
    "bpx_conv1x1_c1_wgrad": ([_i, _i64, _vp, Tensor, _vp, _vp, _i64, _vp], _i),
    "bpx_cast": ([_i, _vp, _i, _vp, _i64, _vp], _i),
    "bpx_upsample_c1_fwd": ([_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "bpx_upsample_c1_blocks": ([_i64], _i),
    "bpx_upsample_c1_bwd": ([_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp], _i),
}

EXPORTS = tuple(_SIGS.keys())


class BpxError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is the product and there is no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C biapy_amd/csrc`."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError = the library does not export what the header declares
        fn.argtypes = args
        fn.restype = res
    # A/B hooks through the environment: wgrad partial-slab cap in percent
    for env, hook in (("BPX_WGRAD_CAP", "bpx_debug_set_wgrad_cap"), ("BPX_WGRAD_K1", "bpx_debug_set_wgrad_k1"),
                      ("BPX_PW_STREAM", "bpx_debug_set_pw_stream"), ("BPX_C1_PERSIST", "bpx_debug_set_c1_persist"),
                      ("BPX_BWD_RS", "bpx_debug_set_bwd_rs")):
        if os.environ.get(env) is not None:
            getattr(lib, hook)(int(os.environ[env]))
    return lib


class PackJob(C.Structure):
    """bpx_pack_job (include/biapy_amd.h)."""
    _fields_ = [("w_d", C.c_void_p), ("packed_d", C.c_void_p), ("mode", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("reserved", C.c_int32)]


class AdamTensor(C.Structure):
    """bpx_adam_tensor (include/biapy_amd.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p), ("numel", C.c_int64)]


class Profile:
    """Per-launch HIP-event timing of selected C-ABI calls (events are recorded on the stream the kernels
    run on - PyTorch's current stream).  Used by bench.py for the roofline figure and the breakdown."""

    def __init__(self, names=None):
        self.names = None if names is None else set(names)
        self.records = []  # (name, int args, start event, end event)

    def want(self, name: str) -> bool:
        return self.names is None or name in self.names

    def add(self, name, args, e0, e1):
        key = []
        for a in args[:-1]:  # the last argument of every entry point is the stream
            if isinstance(a, Tensor):
                key.append("C%d" % a.C)
            elif isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 20):
                key.append(a)
        self.records.append((name, tuple(key), e0, e1))

    def summary(self):
        """{(name, small-int args): [count, total ms]} - call after torch.cuda.synchronize()."""
        out = {}
        for name, args, e0, e1 in self.records:
            k = (name, args)
            c = out.setdefault(k, [0, 0.0])
            c[0] += 1
            c[1] += e0.elapsed_time(e1)
        return out


class _LibProxy:
    """Attribute-compatible wrapper of the CDLL that can time calls when ``prof`` is set."""

    def __init__(self, raw):
        object.__setattr__(self, "_raw", raw)
        object.__setattr__(self, "prof", None)

    def __getattr__(self, name):
        fn = getattr(self._raw, name)

        def call(*a):
            p = self.prof
            if p is None or not p.want(name):
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            p.add(name, a, e0, e1)
            return rc

        object.__setattr__(self, name, call)
        return call


lib = _LibProxy(_load())


def source_digest(only=None) -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, *.cpp in name order): stamped into the PMC summaries under profiles/ so that bench.py
    can tell when a committed counter file no longer belongs to the kernels it is quoted for.  ``only``: file names to restrict it to (the
    blend / gather counters depend on tiling.hip and bpx_common.h alone)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(_HERE, "csrc", "*.cpp"))):
        if only is not None and os.path.basename(f) not in only:
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def check(rc: int) -> None:
    if rc != 0:
        raise BpxError(lib.bpx_last_error().decode())


def stream_ptr() -> Optional[int]:
    """The current PyTorch HIP stream of the calling thread (kernels are launched on it)."""
    return torch.cuda.current_stream().cuda_stream


def dt_of(t: torch.Tensor) -> int:
    return {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16, torch.uint8: U8}[t.dtype]


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


PLANE_PAD_BYTES = int(os.environ.get("BPX_PLANE_PAD", "0"))   # test hook (multiple of 16): extra distance between the planes of a Planar buffer


class Planar:
    """Chunk-planar NDHWC buffer (``bpx_tensor.cs != 0``): the 16-channel chunk ``k`` of every voxel lives in plane ``k`` of a
    ``(C/16, B, D, H, W, 16)`` tensor.  The decoder's ``torch.cat([up, skip], 1)`` buffers are kept this way: the transposed conv and the
    encoder write whole planes (full cache lines) instead of 64 of every 96 bytes, pooling reads a dense plane, and the consumers,
    which walk their input in 16-channel chunks anyway, add a plane stride instead of 32 bytes per chunk."""

    def __init__(self, B: int, S, C: int, dtype, device):
        assert C % 16 == 0
        n = B * 16
        for v in S:
            n *= v
        es = torch.empty((), dtype=dtype).element_size()
        # planes of a 128^3 x 4 patch batch are exactly 2^28 bytes apart; padding them (256 B ... 1 MB, BPX_PLANE_PAD) was measured and
        # changes nothing - no channel / bank aliasing between the planes of a voxel
        self.plane = n + PLANE_PAD_BYTES // es             # elements between consecutive chunks (bpx_tensor.cs)
        self._flat = torch.empty((C // 16) * self.plane, dtype=dtype, device=device)
        self.t = self._flat.as_strided((C // 16, B) + tuple(S) + (16,), (self.plane,) + torch.empty((B,) + tuple(S) + (16,), device="meta").stride())
        self.shape = (B,) + tuple(S) + (C,)
        self.device, self.dtype = self.t.device, dtype

    def dense(self) -> torch.Tensor:
        """The same data as an ordinary (B, D, H, W, C) tensor (tests, debugging)."""
        n = self.t.dim()
        return self.t.permute(*range(1, n - 1), 0, n - 1).reshape(self.shape).contiguous()

    def copy_from_dense(self, x: torch.Tensor) -> "Planar":
        n = self.t.dim()
        self.t.copy_(x.reshape(self.shape[:-1] + (self.shape[-1] // 16, 16)).permute(n - 2, *range(0, n - 2), n - 1))
        return self


def tview(t, c0: int = 0, c: Optional[int] = None) -> Tensor:
    """bpx_tensor for an NDHWC torch tensor (last dim = channels, contiguous) or a ``Planar`` buffer, optionally a channel slice."""
    if t is None:
        return NULL_T
    if isinstance(t, Planar):
        C = t.shape[-1]
        cc = C - c0 if c is None else c
        assert c0 % 16 == 0 and cc % 16 == 0 and c0 + cc <= C, "slices of a chunk-planar buffer are whole 16-channel chunks"
        return Tensor(t.t.data_ptr() + (c0 // 16) * t.plane * t.t.element_size(), 16, cc, t.plane)
    assert t.is_contiguous(), "NDHWC buffers must be contiguous"
    ld = t.shape[-1]
    cc = ld - c0 if c is None else c
    return Tensor(t.data_ptr() + c0 * t.element_size(), ld, cc, 0)
