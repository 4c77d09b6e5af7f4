"""Per-phase cycle anatomy of the ROLE-SPLIT fused conv-backward kernel (bwd_fused.hip, conv3_bwd_rs_kernel) from its in-kernel cycle stamps
(first D wave and first W wave of each workgroup, 5th tile).  Needs a -DBPX_BWD_STAMPS build (bash scripts/ab_build_flags.sh stamps -DBPX_BWD_STAMPS).

    BPX_LIB_PATH=biapy_amd/libbiapy_amd_stamps.so python scripts/bwd_rs_stamps.py [S Ct]      default: 128 48
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from biapy_amd import _lib as L

lib = L.lib
S, ct = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (128, 48)
B, cdy = 4, 16
lib.bpx_debug_set_bwd_rs(3)
st = L.stream_ptr()
dy = torch.randn(B, S, S, S, cdy, device="cuda").to(torch.bfloat16)
tt = torch.randn(B, S, S, S, ct, device="cuda").to(torch.float16)
tp = L.Planar(B, (S, S, S), ct, torch.float16, "cuda").copy_from_dense(tt) if ct > 16 else None
tv = L.tview(tp) if ct > 16 else L.tview(tt)
g = torch.empty(B, S, S, S, ct, device="cuda", dtype=torch.bfloat16)
n = lib.bpx_packed_weight_elems(L.PK_K3_T, ct, cdy, L.MIX16)
wpt = torch.empty(n, dtype=torch.bfloat16, device="cuda")
L.check(lib.bpx_pack_weight(L.PK_K3_T, (torch.randn(cdy, ct, 3, 3, 3, device="cuda") * 0.05).data_ptr(), ct, cdy, L.MIX16, wpt.data_ptr(), st))
rec = torch.rand(B, ct, 4, device="cuda")
red = torch.empty(B, lib.bpx_conv3d_bwd_fused_stats_tiles(B, S, S, S, ct, cdy), 2, ct, device="cuda")
dw = torch.empty(cdy, ct, 3, 3, 3, device="cuda"); db = torch.zeros(cdy, device="cuda")
ws = torch.empty(lib.bpx_conv3d_bwd_fused_workspace(B, S, S, S, ct, cdy), dtype=torch.uint8, device="cuda")
stamps = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")


def run():
    L.check(lib.bpx_conv3d_bwd_fused(L.MIX16, B, S, S, S, L.tview(dy), wpt.data_ptr(), tv, rec.data_ptr(), 1, L.tview(g), red.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                     None, ws.data_ptr(), ws.numel(), st))


run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
print(f"bwd_fused role-split {S}^3 dy {cdy} -> g {ct}: {e0.elapsed_time(e1) * 100:.1f} us per launch (incl. the partial reduction)")
lib.bpx_debug_set_conv_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize()
lib.bpx_debug_set_conv_stamps(None)
s = stamps.cpu().numpy()
s = s[s[:, 0] != 0]
print(f"{len(s)} workgroups; counter ticks (median / mean / p90) per phase")
for role, off, names in (("D wave 0", 0, ["wait at the tile barrier", "dgrad MFMA steps", "dgrad epilogue (ELU', per-lane stats, stores)"]),
                         ("W wave 4", 8, ["own pieces landed (vmcnt 0), A free (counter)", "transform t -> act(t) bf16", "wait at the tile barrier", "next tile set up (bases)",
                                          "next tile's 12 / 8 LDS-DMA pieces requested", "wgrad MFMA steps"])):
    r = s[:, off:off + len(names) + 1]
    d = np.diff(r, axis=1).astype(np.float64)
    print(f" {role}")
    for i, nm in enumerate(names):
        print(f"  {nm:48s} {np.median(d[:, i]):9.0f} {d[:, i].mean():9.0f} {np.percentile(d[:, i], 90):9.0f}")
    tot = (r[:, -1] - r[:, 0]).astype(np.float64)
    print(f"  {'tile total':48s} {np.median(tot):9.0f} {tot.mean():9.0f} {np.percentile(tot, 90):9.0f}")
