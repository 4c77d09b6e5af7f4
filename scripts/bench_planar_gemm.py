"""A/B of the fused shortcut-dgrad / norm-backward GEMM (bpx_conv1x1_fwd_split) and the transposed conv with interleaved vs chunk-planar
concat buffers, standalone (cfg-2 level-0 shapes).  python scripts/bench_planar_gemm.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from biapy_amd import _lib as L
lib = L.lib
DEV = "cuda"
dt, T = L.BF16, torch.bfloat16
B, S = 4, (128, 128, 128)
vox = S[0] * S[1] * S[2]
st = L.stream_ptr()

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def pack(w, mode, cin, cout):
    out = torch.empty(lib.bpx_packed_weight_elems(mode, cin, cout, dt), dtype=T, device=DEV)
    L.check(lib.bpx_pack_weight(mode, w.data_ptr(), cin, cout, dt, out.data_ptr(), st))
    return out

dy = torch.randn(B, *S, 16, device=DEV).to(T)
g0 = torch.randn(B, *S, 48, device=DEV).to(T)
td = torch.randn(B, *S, 48, device=DEV).to(T)
tp = L.Planar(B, S, 48, T, DEV).copy_from_dense(td)
wsct = pack(torch.randn(16, 48, 1, 1, 1, device=DEV) * 0.1, L.PK_DENSE_T, 48, 16)
coef = torch.randn(B, 48, 4, device=DEV)
lo = torch.empty(B, *S, 32, dtype=T, device=DEV); hi = torch.empty(B, *S, 16, dtype=T, device=DEV)
for name, t in (("interleaved", td), ("planar", tp)):
    ms = timeit(lambda: L.check(lib.bpx_conv1x1_fwd_split(dt, B, vox, L.tview(dy), wsct.data_ptr(), None, L.tview(g0), L.tview(t), coef.data_ptr(), L.NULL_T,
                                                          L.tview(lo), L.tview(hi), st)))
    print(f"conv1x1_fwd_split t {name:12s}: {ms * 1e3:8.1f} us")
# transposed conv 32 -> 32, 64^3 -> 128^3, into channels [0, 32) of the 48-channel concat buffer
Sl = (64, 64, 64)
xl = torch.randn(B, *Sl, 32, device=DEV).to(T)
wT = pack(torch.randn(32, 32, 2, 2, 2, device=DEV) * 0.1, L.PK_CT, 32, 32)
bT = torch.randn(32, device=DEV)
ct = lib.bpx_convT3d_stats_tiles(*Sl, 2)
part = torch.zeros(B, ct, 2, 32, device=DEV)
for name, cat in (("interleaved", torch.empty(B, *S, 48, dtype=T, device=DEV)), ("planar", L.Planar(B, S, 48, T, DEV)), ("dense32", torch.empty(B, *S, 32, dtype=T, device=DEV))):
    ms = timeit(lambda: L.check(lib.bpx_convT3d_k2s2_fwd(dt, B, *Sl, 2, L.tview(xl), wT.data_ptr(), bT.data_ptr(), L.tview(cat, 0, 32), part.data_ptr(), st)))
    print(f"convT fwd -> {name:12s}: {ms * 1e3:8.1f} us   ({(xl.numel() + B * vox * 32) * 2 / ms / 1e6:7.1f} GB/s)")
