"""Per-phase cycle anatomy of the z-marching forward kernel (conv3d_zmarch.hip) from in-kernel cycle stamps of every workgroup's 5th step.

    bash scripts/ab_build_flags.sh zmstamps -DBPX_ZM_STAMPS          (the stamps are compiled into a profiling build only)
    BPX_LIB_PATH=$PWD/biapy_amd/libbiapy_amd_zmstamps.so BPX_CONV_ZM=2 python scripts/zm_stamps.py [layer index in tests/bench_kernels.FWD_LAYERS: 0 = 48 -> 16, 1 = 16 -> 16 + sc48, 2 = 16 -> 16 + image]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from biapy_amd import _lib as L
from bench_kernels import FWD_LAYERS

lib = L.lib
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
S, cin, cout, csc = FWD_LAYERS[idx]
B, dt, T = 4, L.F16, torch.float16
st = L.stream_ptr()
x = torch.randn(B, S, S, S, cin, device="cuda").to(T)
if cin == 48:
    x = L.Planar(B, (S, S, S), cin, T, "cuda").copy_from_dense(x)
y = torch.empty(B, S, S, S, cout, device="cuda", dtype=T)


def pack(w, mode, ci, co):
    out = torch.empty(lib.bpx_packed_weight_elems(mode, ci, co, dt), dtype=T, device="cuda")
    L.check(lib.bpx_pack_weight(mode, w.data_ptr(), ci, co, dt, out.data_ptr(), st))
    return out


wp = pack(torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05, L.PK_K3, cin, cout)
bias = torch.zeros(cout, device="cuda")
rec = torch.rand(B, cin, 4, device="cuda")
tiles = lib.bpx_conv3d_stats_tiles(dt, B, S, S, S, cout)
part = torch.empty(B, tiles, 2, cout, device="cuda")
sct, wscp, keep = L.NULL_T, None, []
if csc == 1:
    img = torch.randn(B, S, S, S, device="cuda"); wsc = torch.randn(cout, device="cuda")
    sct, wscp, keep = L.Tensor(img.data_ptr(), 1, 1), wsc.data_ptr(), [img, wsc]
elif csc:
    sc = L.Planar(B, (S, S, S), csc, T, "cuda").copy_from_dense(torch.randn(B, S, S, S, csc, device="cuda").to(T))
    wk = pack(torch.randn(cout, csc, 1, 1, 1, device="cuda"), L.PK_K1, csc, cout)
    sct, wscp, keep = L.tview(sc), wk.data_ptr(), [sc, wk]
stamps = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")


def run():
    L.check(lib.bpx_conv3d_fwd(dt, B, S, S, S, L.tview(x), rec.data_ptr(), 1, wp.data_ptr(), bias.data_ptr(), sct, wscp, bias.data_ptr() if csc else None,
                               L.tview(y), part.data_ptr(), st))


run(); torch.cuda.synchronize()
n0 = lib.bpx_debug_conv_zm_launches()
lib.bpx_debug_set_conv_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize()
lib.bpx_debug_set_conv_stamps(None)
assert lib.bpx_debug_conv_zm_launches() == n0 + 1, "the launch did not take the z-marching kernel"
s = stamps.cpu().numpy()
split = os.environ.get("BPX_STAMP_SPLIT") is not None       # a library built with -DBPX_ZM_SPLIT=<stores after the prefetch> has one more stamp
s = s[s[:, 0] != 0][:, :11 if split else 10]
d = np.diff(s, axis=1).astype(np.float64)
names = (["chunk 0: wait for the prefetched pieces", "chunk 0: transform + LDS write"] if split else ["chunk 0: wait prefetch + transform + LDS write"]) + [ "barrier (planes in LDS)", "request next stage's planes (+ image)", "chunk 0: 14 MFMA steps",
         "chunks 1.. (barrier, transform, barrier, requests, MFMA steps)", "wide shortcut K steps", "epilogue (wait, math, stores, statistics to LDS)",
         "closing barrier", "statistics row"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f"z-march {S}^3 {cin}->{cout} sc={csc} (fp16, B=4): {len(s)} workgroups stamped, {e0.elapsed_time(e1) * 1e3:.0f} us per launch; cycles per phase of a steady-state step "
      f"(median / mean / p90), wave 0")
for i in range(d.shape[1]):
    print(f"  {names[i]:64s} {np.median(d[:, i]):8.0f} {d[:, i].mean():8.0f} {np.percentile(d[:, i], 90):8.0f}")
tot = (s[:, -1] - s[:, 0]).astype(np.float64)
print(f"  {'step total':64s} {np.median(tot):8.0f} {tot.mean():8.0f} {np.percentile(tot, 90):8.0f}")
