"""Per-phase cycle anatomy of the plain conv kernel from in-kernel cycle stamps (profiling aid).

    python scripts/conv_stamps.py [layer index in tests/bench_kernels.FWD_LAYERS]
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
B, dt, T = 4, L.BF16, torch.bfloat16
st = L.stream_ptr()
x = torch.randn(B, S, S, S, cin, device="cuda").to(T)
y = torch.empty(B, S, S, S, cout, device="cuda", dtype=T)
w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
n = lib.bpx_packed_weight_elems(L.PK_K3, cin, cout, dt)
wp = torch.empty(n, dtype=T, device="cuda")
L.check(lib.bpx_pack_weight(L.PK_K3, w.data_ptr(), cin, cout, dt, wp.data_ptr(), st))
bias = torch.zeros(cout, device="cuda")
rec = torch.rand(B, cin, 4, device="cuda")
tiles = lib.bpx_conv3d_stats_tiles(dt, B, S, S, S, cout)
part = torch.empty(B, tiles, 2, cout, device="cuda")
nblk = B * tiles
if os.environ.get('BPX_STAMP_NAMES') == 'persist':
    nblk = 512
stamps = torch.zeros(max(nblk, 4096), 16, dtype=torch.int64, device="cuda")


def run():
    L.check(lib.bpx_conv3d_fwd(dt, B, S, S, S, L.tview(x), rec.data_ptr(), 1, wp.data_ptr(), bias.data_ptr(), L.NULL_T, None, None, L.tview(y),
                               part.data_ptr(), st))


if os.environ.get("BPX_WS") is not None:
    lib.bpx_debug_set_conv_ws(int(os.environ["BPX_WS"]))
run(); torch.cuda.synchronize()
lib.bpx_debug_set_conv_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize()
lib.bpx_debug_set_conv_stamps(None)
s = stamps.cpu().numpy()
s = s[s[:, 0] != 0]
nblk = len(s)
hw = s[:, 15].copy()
s = s[:, :15]
nz = (s != 0).sum(1).max()
d = np.diff(s[:, :nz], axis=1).astype(np.float64)
if os.environ.get("BPX_STAMP_NAMES") == "persist":
    names = ["prologue (load+transform stage 0, load stage 1)"]
    k = 0
    while len(names) < nz:
        names += [f"stage{k} bookkeeping+index math", f"stage{k} step loop", f"stage{k} epilogue", f"stage{k} barrier"]
        k += 1
elif os.environ.get("BPX_WS", "0") in ("0", "5") and S >= 64:
    names = ["tile decode + chunk0 load+transform+write", "barrier", "chunk0 step loop", "chunks 1..", "shortcut + epilogue", "stats reduction + store"]
    nblk = None
elif os.environ.get("BPX_WS", "0") == "3" and S >= 64:
    names = ["index math", "chunk0 load+transform+write", "barrier0", "chunk0 step loop"]
    names += [f"chunk{k} barrier+stage+barrier+step loop" for k in range(1, nz - 5)] + ["final barrier + shortcut + epilogue"]
else:
    names = ["index math", "chunk0 load+transform+write", "barrier0"]
    k = 0
    while len(names) < nz - 2:
        names += [f"chunk{k} step loop", f"chunk{k} barrier"]
        k += 1
    names = names[: nz - 2] + ["epilogue"]
print(f"layer {S}^3 {cin}->{cout}: {nblk} workgroups, {nz} stamps; cycles (median / mean / p90) per phase, wave 0 of each workgroup")
for i in range(d.shape[1]):
    print(f"  {names[i] if i < len(names) else str(i):32s} {np.median(d[:, i]):9.0f} {d[:, i].mean():9.0f} {np.percentile(d[:, i], 90):9.0f}")
tot = (s[:, nz - 1] - s[:, 0]).astype(np.float64)
print(f"  {'workgroup total':32s} {np.median(tot):9.0f} {tot.mean():9.0f} {np.percentile(tot, 90):9.0f}")
# residency: workgroups of one CU (XCC, SE, SH, CU ids) share a clock, so their [start, end] intervals can be overlapped
cu = ((hw >> 32) & 0xF) * 65536 + ((hw >> 8) & 0xFF)
occ = []
for c in np.unique(cu):
    m = cu == c
    st_, en_ = s[m, 0], s[m, nz - 1]
    ev = np.concatenate([np.stack([st_, np.ones_like(st_)], 1), np.stack([en_, -np.ones_like(en_)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    live = np.cumsum(ev[:, 1])
    dtv = np.diff(ev[:, 0])
    occ.append((live[:-1] * dtv).sum() / max(dtv.sum(), 1))
print(f"  {len(np.unique(cu))} distinct CUs; time-averaged resident workgroups per CU: mean {np.mean(occ):.2f}, min {np.min(occ):.2f}, max {np.max(occ):.2f}; "
      f"tiles per CU {np.bincount(np.unique(cu, return_inverse=True)[1]).min()}..{np.bincount(np.unique(cu, return_inverse=True)[1]).max()}")
ok = s[:, 0] > 0
span = s[ok, :nz].max() - s[ok, 0].min()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"  event time {ms * 1e3:.0f} us -> stamp clock = {span / (ms * 1e-3) / 1e6:.0f} MHz")
print(f"  kernel span {span} cycles; sum of workgroup totals / span = {tot.sum() / span:.1f} workgroups in flight (of {256 * 2} slots)")
