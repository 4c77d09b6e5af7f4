"""Where do the z-march and the lean kernel differ?  (debugging aid: prints the first differing voxels of a check_conv3d_zmarch case, twice)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kernel_checks as K
L, lib = K.L, K.lib

def run(f16, B, S, Cin, sc_C, pool, seed=0):
    dt = L.F16 if f16 else L.BF16
    T = K.tdtype(dt)
    D, H, W = S
    g = torch.Generator().manual_seed(seed)
    x = K.rnd(torch.randn(B, D, H, W, Cin, generator=g), dt)
    w = torch.randn(16, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
    bias = (torch.randn(16, generator=g) * 0.1).cuda()
    rec = K.make_recs(B, Cin, seed + 1)[0].cuda()
    wp = K.pack(w, L.PK_K3, Cin, 16, dt)
    xd = K.to_dev(x, dt)
    img = torch.randn(B, D, H, W, generator=g).cuda().contiguous(); wsc = torch.randn(16, generator=g).cuda(); bscd = (torch.randn(16, generator=g) * 0.1).cuda()
    sct, wscp = L.Tensor(img.data_ptr(), 1, 1), wsc.data_ptr()
    tiles = lib.bpx_conv3d_stats_tiles(dt, B, D, H, W, 16)
    outs = []
    for mode in (0, 2, 2):
        lib.bpx_debug_set_conv_zm(mode)
        y = torch.full((B, D, H, W, 16), 7.0, dtype=T, device="cuda")
        part = torch.zeros(B, tiles, 2, 16, device="cuda")
        if pool:
            pooled = torch.full((B, D // pool, H // 2, W // 2, 16), 7.0, dtype=T, device="cuda")
            ppart = torch.zeros(B, tiles, 2, 16, device="cuda")
            L.check(lib.bpx_conv3d_fwd_pool(dt, B, D, H, W, L.tview(xd), L.ptr(rec), 1, wp.data_ptr(), bias.data_ptr(), sct, wscp, L.ptr(bscd), L.tview(y), part.data_ptr(),
                                            pool, L.tview(pooled), ppart.data_ptr(), L.stream_ptr()))
        else:
            L.check(lib.bpx_conv3d_fwd(dt, B, D, H, W, L.tview(xd), L.ptr(rec), 1, wp.data_ptr(), bias.data_ptr(), sct, wscp, L.ptr(bscd), L.tview(y), part.data_ptr(), L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(y.float().cpu())
    lib.bpx_debug_set_conv_zm(-1)
    for k in (1, 2):
        d = (outs[0] != outs[k]).nonzero()
        print(f"case f16={f16} B{B} {S} pool={pool}: run {k}: {len(d)} values differ; first: {d[:24].tolist()}")
        for idx in d[:6].tolist():
            n, z, yy, xx, c = idx
            print("   ", idx, "lean", outs[0][n, z, yy, xx, c].item(), "zm", outs[k][n, z, yy, xx, c].item(), "img", img[n, z, yy, xx].item(), "w1", wsc[c].item())

run(True, 1, (64, 64, 64), 16, 1, 2)
run(True, 3, (33, 41, 49), 16, 1, 0)
run(True, 2, (32, 32, 32), 16, 1, 0)
