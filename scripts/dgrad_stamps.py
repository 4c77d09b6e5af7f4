"""Per-phase cycle anatomy of the lean dgrad kernel from its in-kernel cycle stamps (profiling aid, wave 0 of each workgroup, 5th tile).

    python scripts/dgrad_stamps.py [S cin_of_the_conv cout_of_the_conv]      default: 128 48 16 (dy 16 channels -> g 48 channels)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from biapy_amd import _lib as L

lib = L.lib
S, cin, cout = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 48, 16)
B = 4
mix = os.environ.get("BPX_DT", "mix16") == "mix16"
dt, T = (L.MIX16 if mix else L.BF16), torch.bfloat16
st = L.stream_ptr()
dy = torch.randn(B, S, S, S, cout, device="cuda").to(T)
t = torch.randn(B, S, S, S, cin, device="cuda").to(torch.float16 if mix else T)
g = torch.empty(B, S, S, S, cin, device="cuda", dtype=T)
w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
n = lib.bpx_packed_weight_elems(L.PK_K3_T, cin, cout, L.BF16)
wp = torch.empty(n, dtype=T, device="cuda")
L.check(lib.bpx_pack_weight(L.PK_K3_T, w.data_ptr(), cin, cout, L.BF16, wp.data_ptr(), st))
rec = torch.rand(B, cin, 4, device="cuda")
tiles = lib.bpx_conv3d_stats_tiles(L.BF16, B, S, S, S, cin)
red = torch.empty(B, tiles, 2, cin, device="cuda")
stamps = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")


def run():
    L.check(lib.bpx_conv3d_dgrad(dt, B, S, S, S, L.tview(dy), wp.data_ptr(), L.tview(t), rec.data_ptr(), 1, L.tview(g), red.data_ptr(), st))


run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
print(f"dgrad {S}^3 dy {cout} -> g {cin}: {e0.elapsed_time(e1) * 100:.1f} us per launch")
lib.bpx_debug_set_conv_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize()
lib.bpx_debug_set_conv_stamps(None)
s = stamps.cpu().numpy()
s = s[s[:, 0] != 0][:, :7]
d = np.diff(s, axis=1).astype(np.float64)
names = ["tile decode + chunk0 load + LDS write", "barrier", "chunk0 step loop (14 steps)", "chunks 1..", "epilogue (t loads, ELU', stats, stores)", "stats reduction + store"]
print(f"{len(s)} workgroups; counter ticks (median / mean / p90) per phase")
for i in range(d.shape[1]):
    print(f"  {names[i]:44s} {np.median(d[:, i]):9.0f} {d[:, i].mean():9.0f} {np.percentile(d[:, i], 90):9.0f}")
tot = (s[:, 6] - s[:, 0]).astype(np.float64)
print(f"  {'tile total':44s} {np.median(tot):9.0f} {tot.mean():9.0f} {np.percentile(tot, 90):9.0f}")
