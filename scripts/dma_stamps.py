"""Per-phase cycle anatomy of one steady-state stage of the DMA-pipelined conv kernel (conv3d_dma.hip) from in-kernel cycle stamps.

    python scripts/dma_stamps.py fwd|dgrad <layer index in tests/bench_kernels.FWD_LAYERS>
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
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S, cin, cout, csc = FWD_LAYERS[idx]
B, dt, T = 4, L.BF16, torch.bfloat16
st = L.stream_ptr()


def pack(w, mode_, ci, co):
    n = lib.bpx_packed_weight_elems(mode_, ci, co, dt)
    out = torch.empty(n, dtype=T, device="cuda")
    L.check(lib.bpx_pack_weight(mode_, w.data_ptr(), ci, co, dt, out.data_ptr(), st))
    return out


if mode == "fwd":
    x = torch.randn(B, S, S, S, cin, device="cuda").to(T)
    y = torch.empty(B, S, S, S, cout, device="cuda", dtype=T)
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
        sc = torch.randn(B, S, S, S, csc, device="cuda").to(T)
        wk = pack(torch.randn(cout, csc, 1, 1, 1, device="cuda"), L.PK_K1, csc, cout)
        sct, wscp, keep = L.tview(sc), wk.data_ptr(), [sc, wk]

    def run():
        L.check(lib.bpx_conv3d_fwd(dt, B, S, S, S, L.tview(x), rec.data_ptr(), 1, wp.data_ptr(), bias.data_ptr(), sct, wscp,
                                   bias.data_ptr() if csc else None, L.tview(y), part.data_ptr(), st))
    what = f"fwd {S}^3 {cin}->{cout} sc={csc}"
else:
    dy = torch.randn(B, S, S, S, cout, device="cuda").to(T)
    t = torch.randn(B, S, S, S, cin, device="cuda").to(T)
    g = torch.empty(B, S, S, S, cin, device="cuda", dtype=T)
    wp = pack(torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05, L.PK_K3_T, cin, cout)
    rec = torch.rand(B, cin, 4, device="cuda")
    tiles = lib.bpx_conv3d_stats_tiles(dt, B, S, S, S, cin)
    red = torch.empty(B, tiles, 2, cin, device="cuda")

    def run():
        L.check(lib.bpx_conv3d_dgrad(dt, B, S, S, S, L.tview(dy), wp.data_ptr(), L.tview(t), rec.data_ptr(), 1, L.tview(g), red.data_ptr(), st))
    what = f"dgrad {S}^3 dy{cout}->g{cin}"

stamps = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")
run(); torch.cuda.synchronize()
lib.bpx_debug_set_conv_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize()
lib.bpx_debug_set_conv_stamps(None)
s = stamps.cpu().numpy()
s = s[s[:, 0] != 0][:, :15]
nz = int((s != 0).sum(1).max())
s = s[(s[:, :nz] != 0).all(1)]
d = np.diff(s[:, :nz], axis=1).astype(np.float64)
names = ["weights of the chunk requested + wait for this stage's DMA (vmcnt 0)", "in-place normalise + ELU (LDS -> LDS)", "barrier",
         "request the epilogue operands (t / image / shortcut chunk 0)", "14 MFMA steps + the next stage's DMA pieces", "fused 1x1x1 shortcut (chunks overlapped)", "epilogue math + 16-byte stores", "statistics barrier + store"]
print(f"{what}: {len(s)} workgroups, {nz} stamps; cycles (median / mean / p90) per phase of one steady-state last-chunk stage, wave 0")
for i in range(d.shape[1]):
    print(f"  {names[i] if i < len(names) else str(i):72s} {np.median(d[:, i]):9.0f} {d[:, i].mean():9.0f} {np.percentile(d[:, i], 90):9.0f}")
tot = (s[:, nz - 1] - s[:, 0]).astype(np.float64)
print(f"  {'stage total':72s} {np.median(tot):9.0f} {tot.mean():9.0f} {np.percentile(tot, 90):9.0f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f"  event time {e0.elapsed_time(e1) * 1e3:.0f} us")
