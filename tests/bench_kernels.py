"""Micro-benchmarks of single kernels at the cfg-2 layer shapes (events on the launch stream).

    python tests/bench_kernels.py conv_fwd|conv_dgrad|wgrad|c1|merge|all [--reps 20] [--dtype bf16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from biapy_amd import _lib as L  # noqa: E402

lib = L.lib
DEV = "cuda"
# (spatial, Cin, Cout, shortcut C) of the 3x3x3 convolutions of cfg 2 (B = 4)
FWD_LAYERS = [
    (128, 48, 16, 0), (128, 16, 16, 48), (128, 16, 16, 1), (64, 96, 32, 0), (64, 32, 32, 96), (64, 16, 32, 0), (64, 32, 32, 16),
    (32, 192, 64, 0), (32, 64, 64, 192), (32, 32, 64, 0), (32, 64, 64, 32), (16, 384, 128, 0), (16, 128, 128, 384), (16, 64, 128, 0),
    (16, 128, 128, 64), (8, 128, 256, 0), (8, 256, 256, 128),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="-1", help="index (or comma-separated indices) into the layer table")
    a = ap.parse_args()
    dt = {"bf16": L.BF16, "f16": L.F16}.get(a.dtype, L.F32)       # f16: the forward half of the benched mixed mode
    T = {L.BF16: torch.bfloat16, L.F16: torch.float16}.get(dt, torch.float32)
    B = int(os.environ.get("BPX_BENCH_B", "4"))
    st = L.stream_ptr()
    if os.environ.get('BPX_WS') is not None:
        lib.bpx_debug_set_conv_ws(int(os.environ['BPX_WS']))
    if os.environ.get('BPX_OCC') is not None:
        lib.bpx_debug_set_conv_occ(int(os.environ['BPX_OCC']))

    def pack(w, mode, cin, cout):
        n = lib.bpx_packed_weight_elems(mode, cin, cout, dt)
        out = torch.empty(n, dtype=T, device=DEV)
        L.check(lib.bpx_pack_weight(mode, w.data_ptr(), cin, cout, dt, out.data_ptr(), st))
        return out

    only = [int(v) for v in str(a.only).split(",")]
    layers = FWD_LAYERS if only[0] < 0 else [FWD_LAYERS[k] for k in only]
    if a.what in ("conv_fwd", "all"):
        for (S, cin, cout, csc) in layers:
            x = torch.randn(B, S, S, S, cin, device=DEV).to(T)
            planar = os.environ.get("BPX_BENCH_PLANAR", "1") != "0" and dt != L.F32      # the decoder's concat tensors are chunk-planar in the network
            if planar and cin in (48, 96, 192, 384):
                x = L.Planar(B, (S, S, S), cin, T, DEV).copy_from_dense(x)
            y = torch.empty(B, S, S, S, cout, device=DEV, dtype=T)
            w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
            wp = pack(w, L.PK_K3, cin, cout)
            bias = torch.zeros(cout, device=DEV)
            rec = torch.rand(B, cin, 4, device=DEV)
            tiles = lib.bpx_conv3d_stats_tiles(dt, B, S, S, S, cout)
            part = torch.empty(B, tiles, 2, cout, device=DEV)
            sct, wscp, keep = L.NULL_T, None, []
            if csc == 1:
                img = torch.randn(B, S, S, S, device=DEV); wsc = torch.randn(cout, device=DEV)
                sct, wscp, keep = L.Tensor(img.data_ptr(), 1, 1), wsc.data_ptr(), [img, wsc]
            elif csc:
                sc = torch.randn(B, S, S, S, csc, device=DEV).to(T)
                if planar and csc in (48, 96, 192, 384):
                    sc = L.Planar(B, (S, S, S), csc, T, DEV).copy_from_dense(sc)
                wk = pack(torch.randn(cout, csc, 1, 1, 1, device=DEV), L.PK_K1, csc, cout)
                sct, wscp, keep = L.tview(sc), wk.data_ptr(), [sc, wk]
            recp = None if os.environ.get('BPX_NO_NORM') else rec.data_ptr()   # ablation: skip the fused normalise+ELU prologue
            f = lambda: L.check(lib.bpx_conv3d_fwd(dt, B, S, S, S, L.tview(x), recp, 1, wp.data_ptr(), bias.data_ptr(), sct, wscp,
                                                   bias.data_ptr() if csc else None, L.tview(y), part.data_ptr(), st))
            ms = timeit(f, a.reps)
            fl = 2 * B * S ** 3 * (27 * cin + csc) * cout
            by = B * S ** 3 * (cin + cout + csc) * y.element_size()
            print(f"conv_fwd  {S:4d}^3 {cin:4d}->{cout:4d} sc={csc:4d}: {ms * 1e3:9.1f} us {fl / ms / 1e9:8.1f} TF/s  {by / ms / 1e6:8.1f} GB/s(alg)")
    if a.what in ("conv_dgrad", "all"):
        for (S, cin, cout, csc) in layers:
            dy = torch.randn(B, S, S, S, cout, device=DEV).to(T)
            t = torch.randn(B, S, S, S, cin, device=DEV).to(T)
            g = torch.empty(B, S, S, S, cin, device=DEV, dtype=T)
            wp = pack(torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05, L.PK_K3_T, cin, cout)
            rec = torch.rand(B, cin, 4, device=DEV)
            tiles = lib.bpx_conv3d_stats_tiles(dt, B, S, S, S, cin)
            red = torch.empty(B, tiles, 2, cin, device=DEV)
            f = lambda: L.check(lib.bpx_conv3d_dgrad(dt, B, S, S, S, L.tview(dy), wp.data_ptr(), L.tview(t), rec.data_ptr(), 1, L.tview(g),
                                                     red.data_ptr(), st))
            ms = timeit(f, a.reps)
            fl = 2 * B * S ** 3 * 27 * cin * cout
            print(f"conv_dgrad {S:4d}^3 dy{cout:4d}->g{cin:4d}: {ms * 1e3:9.1f} us {fl / ms / 1e9:8.1f} TF/s")
    if a.what == "stream":
        # the materialised streaming passes of the cfg-2 backward at level 0 (mixed mode: fp16 activations, bf16 gradients), GB/s of algorithmic traffic
        S, C = 128, 16
        vox = S ** 3
        gq = torch.randn(B, S, S, S, C, device=DEV).to(torch.bfloat16)
        tq = torch.randn(B, S, S, S, C, device=DEV).to(torch.float16)
        out = torch.empty_like(gq)
        coef = torch.rand(B, C, 4, device=DEV)
        f = lambda: L.check(lib.bpx_norm_bwd_apply(L.MIX16, B, vox, L.tview(gq), L.tview(tq), coef.data_ptr(), L.NULL_T, L.tview(out), st))
        ms = timeit(f, a.reps)
        print(f"norm_bwd_apply {S}^3 C{C}: {ms * 1e3:8.1f} us  {3 * B * vox * C * 2 / ms / 1e6:8.1f} GB/s")
        f = lambda: L.check(lib.bpx_norm_bwd_apply(L.MIX16, B, vox, L.tview(gq), L.tview(tq), coef.data_ptr(), L.tview(out), L.tview(out), st))
        ms = timeit(f, a.reps)
        print(f"norm_bwd_apply {S}^3 C{C} + addend: {ms * 1e3:8.1f} us  {4 * B * vox * C * 2 / ms / 1e6:8.1f} GB/s")
        w = torch.randn(1, C, device=DEV)
        dl = torch.randn(B, 1, S, S, S, device=DEV)
        dw, db = torch.zeros(1, C, device=DEV), torch.zeros(1, device=DEV)
        ws = torch.empty(lib.bpx_head_bwd_workspace(C, 1), dtype=torch.uint8, device=DEV)
        f = lambda: L.check(lib.bpx_head_bwd(L.MIX16, vox, B, L.tview(tq), w.data_ptr(), 1, dl.data_ptr(), vox, vox, L.tview(out), dw.data_ptr(), db.data_ptr(),
                                             ws.data_ptr(), ws.numel(), st))
        ms = timeit(f, a.reps)
        print(f"head_bwd {S}^3 C{C} -> 1: {ms * 1e3:8.1f} us  {B * vox * (2 * C * 2 + 4) / ms / 1e6:8.1f} GB/s")
        return
    if a.what == "k1":
        # k = 1 weight gradients of raw inputs (the blocks' shortcuts) in the mixed mode: streaming kernel vs the generic tile kernel
        for (S, cin, cout, planar) in [(128, 48, 16, True), (64, 96, 32, True), (64, 16, 32, False)]:
            xx = torch.randn(B, S, S, S, cin, device=DEV).to(torch.float16)
            xp = L.Planar(B, (S, S, S), cin, torch.float16, DEV).copy_from_dense(xx) if planar else None
            xv = L.tview(xp) if planar else L.tview(xx)
            dy = torch.randn(B, S, S, S, cout, device=DEV).to(torch.bfloat16)
            dw = torch.empty(cout, cin, 1, 1, 1, device=DEV)
            ws = torch.empty(max(1, lib.bpx_conv3d_wgrad_workspace(B, S, S, S, cin, cout, 1)), dtype=torch.uint8, device=DEV)
            f = lambda: L.check(lib.bpx_conv3d_wgrad(L.MIX16, B, S, S, S, xv, None, 0, L.tview(dy), 1, dw.data_ptr(), None, ws.data_ptr(), ws.numel(), st))
            by = B * S ** 3 * (cin + cout) * 2
            for on in (0, 1):
                lib.bpx_debug_set_wgrad_k1(on)
                ms = timeit(f, a.reps)
                print(f"wgrad k=1 {S:4d}^3 {cin:3d}.{cout:3d} {'stream' if on else 'tile  '}: {ms * 1e3:8.1f} us (+ reduce) {by / ms / 1e9:7.2f} TB/s")
        return
    if a.what == "pws":
        # the decoder blocks' input gradient (bpx_conv1x1_fwd_split + IN-backward affine, mixed mode): streaming kernel vs the tile kernel
        for (S, K) in [(128, 16), (64, 32)]:
            vps = S ** 3
            C3 = 3 * K
            xx = torch.randn(B, vps, K, device=DEV).to(torch.bfloat16)
            gg = torch.randn(B, vps, C3, device=DEV).to(torch.bfloat16)
            tt = torch.randn(B, S, S, S, C3, device=DEV).to(torch.float16)
            tp = L.Planar(B, (S, S, S), C3, torch.float16, DEV).copy_from_dense(tt)
            coef = torch.randn(B, C3, 4, device=DEV)
            wp = pack(torch.randn(C3, K, 1, 1, 1, device=DEV), L.PK_DENSE, K, C3)
            y_lo = torch.empty(B, vps, 2 * K, device=DEV, dtype=torch.bfloat16); y_hi = torch.empty(B, vps, K, device=DEV, dtype=torch.bfloat16)
            f = lambda: L.check(lib.bpx_conv1x1_fwd_split(L.MIX16, B, vps, L.tview(xx), wp.data_ptr(), None, L.tview(gg), L.tview(tp), coef.data_ptr(), L.NULL_T,
                                                          L.tview(y_lo), L.tview(y_hi), st))
            by = B * vps * (K + 3 * C3) * 2
            for on in (0, 1):
                lib.bpx_debug_set_pw_stream(on)
                ms = timeit(f, a.reps)
                print(f"1x1x1 + IN-backward {S:4d}^3 {K:3d} -> {C3:3d} {'stream' if on else 'tile  '}: {ms * 1e3:8.1f} us {by / ms / 1e9:7.2f} TB/s")
        return
    if a.what == "prologue":
        # VERDICT r3 next #5 ("forward prologue diet"): level-0 forward convs with the fused normalise + ELU prologue, against a streaming pass
        # that materialises a = ELU(scale * x + shift) once (bpx_norm_act_fwd) followed by the same conv WITHOUT a prologue.
        for (S, cin, cout, csc) in [(128, 16, 16, 48), (128, 16, 16, 1), (128, 48, 16, 0), (64, 32, 32, 96), (64, 96, 32, 0)]:
            x = torch.randn(B, S, S, S, cin, device=DEV).to(T)
            av = torch.empty_like(x)
            y = torch.empty(B, S, S, S, cout, device=DEV, dtype=T)
            wp = pack(torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05, L.PK_K3, cin, cout)
            bias = torch.zeros(cout, device=DEV)
            rec = torch.rand(B, cin, 4, device=DEV)
            tiles = lib.bpx_conv3d_stats_tiles(dt, B, S, S, S, cout)
            part = torch.empty(B, tiles, 2, cout, device=DEV)
            sct, wscp, keep = L.NULL_T, None, []
            if csc == 1:
                img = torch.randn(B, S, S, S, device=DEV); wsc = torch.randn(cout, device=DEV)
                sct, wscp, keep = L.Tensor(img.data_ptr(), 1, 1), wsc.data_ptr(), [img, wsc]
            elif csc:
                sc = torch.randn(B, S, S, S, csc, device=DEV).to(T)
                if planar and csc in (48, 96, 192, 384):
                    sc = L.Planar(B, (S, S, S), csc, T, DEV).copy_from_dense(sc)
                wk = pack(torch.randn(cout, csc, 1, 1, 1, device=DEV), L.PK_K1, csc, cout)
                sct, wscp, keep = L.tview(sc), wk.data_ptr(), [sc, wk]

            def conv(src, recp):
                return lambda: L.check(lib.bpx_conv3d_fwd(dt, B, S, S, S, L.tview(src), recp, 1, wp.data_ptr(), bias.data_ptr(), sct, wscp,
                                                          bias.data_ptr() if csc else None, L.tview(y), part.data_ptr(), st))
            fpass = lambda: L.check(lib.bpx_norm_act_fwd(dt, B, S ** 3, L.tview(x), rec.data_ptr(), 1, L.tview(av), st))
            m_fused, m_plain, m_pass = timeit(conv(x, rec.data_ptr()), a.reps), timeit(conv(av, None), a.reps), timeit(fpass, a.reps)
            print(f"prologue {S:4d}^3 {cin:3d}->{cout:3d} sc={csc:3d}: fused prologue {m_fused * 1e3:7.1f} us | materialise {m_pass * 1e3:7.1f} + conv without prologue "
                  f"{m_plain * 1e3:7.1f} = {(m_pass + m_plain) * 1e3:7.1f} us")
        return
    if os.environ.get('BPX_TILE_ORDER') is not None:
        lib.bpx_debug_set_tile_order(int(os.environ['BPX_TILE_ORDER']))
    if os.environ.get('BPX_WGRAD') is not None:
        lib.bpx_debug_set_wgrad_tr(int(os.environ['BPX_WGRAD']))   # 1 auto, 3 never shift-dy, 5 always
    if a.what in ("wgrad", "all"):
        for (S, cin, cout, csc) in layers:
            x = torch.randn(B, S, S, S, cin, device=DEV).to(T)
            dy = torch.randn(B, S, S, S, cout, device=DEV).to(T)
            rec = torch.rand(B, cin, 4, device=DEV)
            dw = torch.empty(cout, cin, 3, 3, 3, device=DEV)
            db = torch.zeros(cout, device=DEV)
            ws = torch.empty(max(1, lib.bpx_conv3d_wgrad_workspace(B, S, S, S, cin, cout, 3)), dtype=torch.uint8, device=DEV)
            f = lambda: L.check(lib.bpx_conv3d_wgrad(dt, B, S, S, S, L.tview(x), rec.data_ptr(), 1, L.tview(dy), 3, dw.data_ptr(), db.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), st))
            ms = timeit(f, a.reps)
            fl = 2 * B * S ** 3 * 27 * cin * cout
            print(f"wgrad     {S:4d}^3 {cin:4d}->{cout:4d}: {ms * 1e3:9.1f} us {fl / ms / 1e9:8.1f} TF/s  ws={ws.numel() / 1e6:.1f} MB")
    if a.what in ("bwd", "all"):
        # backward of one conv in the mixed training mode (t fp16, gradients bf16): dgrad + wgrad as two kernels vs bpx_conv3d_bwd_fused.
        # (S, Ct = channels of t / g, Cdy, planar t): the level-0 convs of cfg 2 (decoder conv2 / encoder conv2, decoder conv1)
        for (S, ct, cdy, planar) in [(128, 16, 16, False), (128, 48, 16, True), (64, 16, 16, False), (64, 32, 32, False), (64, 96, 32, True), (64, 16, 32, False)]:
            dy = torch.randn(B, S, S, S, cdy, device=DEV).to(torch.bfloat16)
            tt = torch.randn(B, S, S, S, ct, device=DEV).to(torch.float16)
            tp = L.Planar(B, (S, S, S), ct, torch.float16, DEV).copy_from_dense(tt) if planar else None   # (kept alive: tview holds only the address)
            tv = L.tview(tp) if planar else L.tview(tt)
            g = torch.empty(B, S, S, S, ct, device=DEV, dtype=torch.bfloat16)
            n = lib.bpx_packed_weight_elems(L.PK_K3_T, ct, cdy, L.MIX16)
            wpt = torch.empty(n, dtype=torch.bfloat16, device=DEV)
            L.check(lib.bpx_pack_weight(L.PK_K3_T, (torch.randn(cdy, ct, 3, 3, 3, device=DEV) * 0.05).data_ptr(), ct, cdy, L.MIX16, wpt.data_ptr(), st))
            rec = torch.rand(B, ct, 4, device=DEV)
            tiles = lib.bpx_conv3d_stats_tiles(L.BF16, B, S, S, S, ct)
            red = torch.empty(B, tiles, 2, ct, device=DEV)
            dw = torch.empty(cdy, ct, 3, 3, 3, device=DEV); db = torch.zeros(cdy, device=DEV)
            ws = torch.empty(max(1, lib.bpx_conv3d_wgrad_workspace(B, S, S, S, ct, cdy, 3)), dtype=torch.uint8, device=DEV)
            fd = lambda: L.check(lib.bpx_conv3d_dgrad(L.MIX16, B, S, S, S, L.tview(dy), wpt.data_ptr(), tv, rec.data_ptr(), 1, L.tview(g), red.data_ptr(), st))
            fw = lambda: L.check(lib.bpx_conv3d_wgrad(L.MIX16, B, S, S, S, tv, rec.data_ptr(), 1, L.tview(dy), 3, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st))
            md, mw = timeit(fd, a.reps), timeit(fw, a.reps)
            U = B * S ** 3 * 32 / 1e6   # MB of one 16-channel tensor
            line = f"bwd {S:4d}^3 dy{cdy}->g{ct:3d}: dgrad {md * 1e3:7.1f} us + wgrad(+reduce) {mw * 1e3:7.1f} us = {(md + mw) * 1e3:7.1f}"
            if lib.bpx_conv3d_bwd_fused_supported(L.MIX16, B, S, S, S, ct, cdy):
                ftiles = lib.bpx_conv3d_bwd_fused_stats_tiles(B, S, S, S, ct, cdy)
                red2 = torch.empty(B, ftiles, 2, ct, device=DEV)
                ws2 = torch.empty(max(1, lib.bpx_conv3d_bwd_fused_workspace(B, S, S, S, ct, cdy)), dtype=torch.uint8, device=DEV)
                ff = lambda: L.check(lib.bpx_conv3d_bwd_fused(L.MIX16, B, S, S, S, L.tview(dy), wpt.data_ptr(), tv, rec.data_ptr(), 1, L.tview(g), red2.data_ptr(),
                                                              dw.data_ptr(), db.data_ptr(), None, ws2.data_ptr(), ws2.numel(), st))
                mf = timeit(ff, a.reps)
                alg = (cdy / 16 + 2 * ct / 16) * U
                line += f" | fused(+reduce) {mf * 1e3:7.1f} us = {alg / mf / 1e3:6.2f} TB/s(alg {alg:.0f} MB), x{(md + mw) / mf:.2f}"
            print(line, flush=True)
    if a.what in ("c1", "all"):
        S = 128
        img = torch.randn(B, S, S, S, device=DEV)
        y = torch.empty(B, S, S, S, 16, device=DEV, dtype=T)
        w = torch.randn(16, 1, 3, 3, 3, device=DEV); b = torch.zeros(16, device=DEV)
        tiles = lib.bpx_conv3d_c1_stats_tiles(S, S, S)
        part = torch.empty(B, tiles, 2, 16, device=DEV)
        ms = timeit(lambda: L.check(lib.bpx_conv3d_c1_fwd(dt, B, S, S, S, img.data_ptr(), w.data_ptr(), b.data_ptr(), L.tview(y), part.data_ptr(), st)), a.reps)
        print(f"c1_fwd 128^3 1->16: {ms * 1e3:9.1f} us  {(B * S ** 3 * (4 + 16 * y.element_size())) / ms / 1e6:8.1f} GB/s(alg)")
        dw = torch.zeros(16, 1, 3, 3, 3, device=DEV); db = torch.zeros(16, device=DEV)
        wsc = torch.empty(lib.bpx_conv3d_c1_wgrad_workspace(16), dtype=torch.uint8, device=DEV)
        ms = timeit(lambda: L.check(lib.bpx_conv3d_c1_wgrad(dt, B, S, S, S, img.data_ptr(), L.tview(y), dw.data_ptr(), db.data_ptr(), wsc.data_ptr(), wsc.numel(), st)), a.reps)
        print(f"c1_wgrad 128^3: {ms * 1e3:9.1f} us")
    if a.what == "merge_rows":      # PMC passes (scripts/refresh_profiles.sh): only the production kernels, whole-volume launches
        from biapy_amd import tiling
        vol = (512, 512, 512)
        plan = tiling.MergePlan(vol, (128, 128, 128), (0.5, 0.5, 0.5), (0, 0, 0), torch.device(DEV))
        patches = torch.rand(plan.n_patches, 128, 128, 128, 1, device=DEV)
        out = torch.empty(vol + (1,), device=DEV)
        v = torch.rand(vol + (1,), device=DEV)
        for _ in range(a.reps):
            tiling.merge_device(patches, plan, out=out)
            tiling.crop_device(v, (128, 128, 128), (0.5, 0.5, 0.5), out=patches)
        torch.cuda.synchronize()
        return
    if a.what in ("merge", "all"):
        from biapy_amd import tiling
        vol = (512, 512, 512)
        plan = tiling.MergePlan(vol, (128, 128, 128), (0.5, 0.5, 0.5), (0, 0, 0), torch.device(DEV))
        patches = torch.rand(plan.n_patches, 128, 128, 128, 1, device=DEV)
        out = torch.empty(vol + (1,), device=DEV)
        by = patches.numel() * 4 + out.numel() * 4
        v = torch.rand(vol + (1,), device=DEV)
        for scalar, tag in ((1, "element-per-thread kernels (round 1)"), (0, "16-byte row kernels")):
            lib.bpx_debug_set_tiling_scalar(scalar)
            ms = timeit(lambda: tiling.merge_device(patches, plan, out=out), max(3, a.reps // 4))
            print(f"merge 512^3 from {plan.n_patches} x 128^3 [{tag}]: {ms:9.3f} ms  {by / ms / 1e6:8.1f} GB/s(alg)  = {by / ms / 1e6 / 8000:.3f} of 8 TB/s")
            ms = timeit(lambda: tiling.crop_device(v, (128, 128, 128), (0.5, 0.5, 0.5), out=patches), max(3, a.reps // 4))
            print(f"crop  512^3 -> {plan.n_patches} x 128^3 [{tag}]: {ms:9.3f} ms  {by / ms / 1e6:8.1f} GB/s(alg)  = {by / ms / 1e6 / 8000:.3f} of 8 TB/s")
        lib.bpx_debug_set_tiling_scalar(0)
        # what the sliding-window predictor launches per batch: 4 patches gathered from the volume
        small = torch.empty((4, 128, 128, 128, 1), device=DEV)
        ms = timeit(lambda: tiling.crop_device(v, (128, 128, 128), (0.5, 0.5, 0.5), c_begin=100, c_count=4, out=small), a.reps)
        print(f"crop  4 x 128^3 (one batch of the predictor): {ms * 1e3:9.1f} us  {2 * small.numel() * 4 / ms / 1e6:8.1f} GB/s(alg)")


def bench_convt(reps=20):
    """Transposed conv forward 32->32 @64^3 -> 128^3 (B=4): output into a channel slice of the 48-channel concat buffer vs a dense tensor."""
    B, S, Cc = 4, 64, 32
    st = L.stream_ptr()
    x = torch.randn(B, S, S, S, Cc, device=DEV).to(torch.bfloat16)
    w = torch.randn(Cc, Cc, 2, 2, 2, device=DEV) / Cc ** 0.5
    b = torch.zeros(Cc, device=DEV)
    n = lib.bpx_packed_weight_elems(L.PK_CT, Cc, Cc, L.BF16)
    wp = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    L.check(lib.bpx_pack_weight(L.PK_CT, w.data_ptr(), Cc, Cc, L.BF16, wp.data_ptr(), st))
    tiles = lib.bpx_convT3d_stats_tiles(S, S, S, 2)
    part = torch.empty(B, tiles, 2, Cc, device=DEV)
    for ld, name in ((48, "slice of a 48-channel buffer"), (32, "dense"), (64, "slice of a 64-channel buffer")):
        y = torch.empty(B, 2 * S, 2 * S, 2 * S, ld, dtype=torch.bfloat16, device=DEV)
        f = lambda: L.check(lib.bpx_convT3d_k2s2_fwd(L.BF16, B, S, S, S, 2, L.tview(x), wp.data_ptr(), b.data_ptr(), L.tview(y, 0, Cc), part.data_ptr(), st))
        ms = timeit(f, reps)
        print(f"convT fwd 32->32 64^3->128^3 into {name}: {ms * 1e3:8.1f} us  {(x.numel() + B * (2 * S) ** 3 * Cc) * 2 / ms / 1e6:8.1f} GB/s")


def bench_rcan():
    """RCAN trunk of cfg 5 (16 filters, 10 groups x 20 RCABs, no up-scaling) on one 64^3 patch: forward and train step (untuned path)."""
    import time

    from biapy_amd.rcan import rcan

    torch.manual_seed(0)
    m = rcan(ndim=3, num_channels=1, filters=16, num_rg=10, num_rcab=20, reduction=16, upscaling_layer=False, out_channels=1, head_activations=["linear"]).cuda()
    x = torch.randn(1, 1, 64, 64, 64, device=DEV)
    t = torch.randn(1, 1, 64, 64, 64, device=DEV)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    with torch.no_grad():
        m.eval()(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / 3
    m.train()
    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.l1_loss(m(x), t)
        loss.backward()
        opt.step()
        return loss
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    tr = (time.perf_counter() - t0) / 3
    fl = 2 * 64 ** 3 * 27 * 16 * 16 * (10 * 20 * 2 + 10 + 1) + 2 * 64 ** 3 * 27 * 16 * 2
    print(f"rcan trunk 64^3 (10x20 RCABs, 16 filters): forward {fwd * 1e3:.1f} ms ({fl / fwd / 1e12:.1f} TFLOP/s), train step {tr * 1e3:.1f} ms "
          f"({64 ** 3 / tr / 1e6:.2f} Mvox/s), loss {loss.item():.4f}", flush=True)
    # the same through HIP-graph replay (the eager numbers are launch-bound: ~1300 / ~5000 launches from Python)
    from biapy_amd.graphs import GraphedInference, GraphedTrainStep

    gi = GraphedInference(m.eval(), x)
    gi()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gi()
    torch.cuda.synchronize()
    gf = (time.perf_counter() - t0) / 5
    print(f"  graph replay: forward {gf * 1e3:.1f} ms ({fl / gf / 1e12:.1f} TFLOP/s)", flush=True)
    # train step replayed from a graph: a FRESH model and a capturable optimizer (capturing it on the model above, after the
    # eager AdamW and the inference graph, dumped core once; in isolation it is fine)
    del gi, opt, m
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    m2 = rcan(ndim=3, num_channels=1, filters=16, num_rg=10, num_rcab=20, reduction=16, upscaling_layer=False, out_channels=1, head_activations=["linear"]).cuda().train()
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-4, capturable=True, fused=True)   # unfused capturable AdamW: two strided divisions per parameter tensor
    gs = GraphedTrainStep(m2, torch.nn.functional.l1_loss, opt2, x, t, warmup=1)
    gs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gs()
    torch.cuda.synchronize()
    gt = (time.perf_counter() - t0) / 5
    print(f"  graph replay: train step {gt * 1e3:.1f} ms ({64 ** 3 / gt / 1e6:.2f} Mvox/s)", flush=True)


def bench_rcan_x4_train():
    """cfg 5 as configured (RCAN x4: 10 x 20 RCABs, 16 filters, 64^3 -> 256^3), TRAIN step in the mixed mode: forward + L1 + backward (the x-scale stage's
    backward = the shuffle's adjoint + wgrad / dgrad per 128-channel block, round 4) + AdamW, eager and replayed from a HIP graph."""
    import time

    from biapy_amd.graphs import GraphedTrainStep
    from biapy_amd.rcan import rcan

    torch.manual_seed(0)
    m = rcan(ndim=3, num_channels=1, filters=16, scale=4, num_rg=10, num_rcab=20, reduction=16, upscaling_layer=True, out_channels=1, head_activations=["linear"],
             compute_dtype=torch.float16).cuda().train()
    x = torch.randn(1, 1, 64, 64, 64, device=DEV)
    t = torch.randn(1, 1, 256, 256, 256, device=DEV)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    for g in (1, 8):
        m.engine().up_group = g
        def step():
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.l1_loss(m(x), t)
            loss.backward()
            opt.step()
            return loss
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            loss = step()
        torch.cuda.synchronize()
        tr = (time.perf_counter() - t0) / 3
        print(f"rcan x4 64^3 -> 256^3 train step, eager, {g} sub-position(s) per backward call: {tr * 1e3:.1f} ms, loss {loss.item():.4f}", flush=True)
    del opt, m                                                              # a fresh model and optimizer for the capture (see bench_rcan)
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    m2 = rcan(ndim=3, num_channels=1, filters=16, scale=4, num_rg=10, num_rcab=20, reduction=16, upscaling_layer=True, out_channels=1, head_activations=["linear"],
              compute_dtype=torch.float16).cuda().train()
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-4, capturable=True, fused=True)
    gs = GraphedTrainStep(m2, torch.nn.functional.l1_loss, opt2, x, t, warmup=1)
    gs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gs()
    torch.cuda.synchronize()
    gt = (time.perf_counter() - t0) / 5
    print(f"  graph replay: train step {gt * 1e3:.1f} ms ({256 ** 3 / gt / 1e6:.1f} M output voxels/s, {64 ** 3 / gt / 1e6:.2f} M input voxels/s)", flush=True)


def bench_rcan_fwd_only():
    """The cfg-5 trunk forward replayed from a HIP graph, nothing else (the command `rocprofv3 --kernel-trace --stats` is run on)."""
    import time

    from biapy_amd.graphs import GraphedInference
    from biapy_amd.rcan import rcan

    torch.manual_seed(0)
    m = rcan(ndim=3, num_channels=1, filters=16, num_rg=10, num_rcab=20, reduction=16, upscaling_layer=False, out_channels=1, head_activations=["linear"],
             compute_dtype=torch.float16).cuda().eval()
    x = torch.randn(1, 1, 64, 64, 64, device=DEV)
    gi = GraphedInference(m, x)
    gi()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gi()
    torch.cuda.synchronize()
    print(f"rcan trunk forward, graph replay: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms", flush=True)


def bench_rcan_train_only():
    """The cfg-5 trunk TRAIN step (mixed mode) replayed from a HIP graph, nothing else (for `rocprofv3 --kernel-trace --stats`)."""
    import time

    from biapy_amd.graphs import GraphedTrainStep
    from biapy_amd.rcan import rcan

    torch.manual_seed(0)
    m = rcan(ndim=3, num_channels=1, filters=16, num_rg=10, num_rcab=20, reduction=16, upscaling_layer=False, out_channels=1, head_activations=["linear"],
             compute_dtype=torch.float16).cuda().train()
    x = torch.randn(1, 1, 64, 64, 64, device=DEV)
    t = torch.randn(1, 1, 64, 64, 64, device=DEV)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, capturable=True, fused=True)
    gs = GraphedTrainStep(m, torch.nn.functional.l1_loss, opt, x, t, warmup=1)
    gs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gs()
    torch.cuda.synchronize()
    print(f"rcan trunk train step (mixed mode), graph replay: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms", flush=True)


def bench_chunked():
    """By-chunks inference of a 512^3 float32 volume: 128^3 patches, padding 16 -> 96^3 chunks (216 of them), cfg-2 ResUNet bf16."""
    import time

    from biapy_amd.chunked import ChunkedPredictor
    from biapy_amd.resunet import ResUNet

    torch.manual_seed(0)
    m = ResUNet(image_shape=(128, 128, 128, 1), activation="elu", feature_maps=[16, 32, 64, 128, 256], drop_values=[0.0] * 5, normalization="in",
                yx_down=[2] * 4, z_down=[2] * 4, isotropy=[True] * 5, larger_io=False, conv_layers=[2] * 5).cuda().eval()
    vol = torch.randn(512, 512, 512, 1, device=DEV)
    pred = ChunkedPredictor(m.predict_proba, (128, 128, 128), (16, 16, 16), batch_size=4)
    pred.predict(vol)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pred.predict(vol)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"by-chunks 512^3, 216 chunks of 96^3 (128^3 patches): {dt * 1e3:.1f} ms  {216 * 128 ** 3 / dt / 1e9:.2f} Gvox/s patch voxels, "
          f"{512 ** 3 / dt / 1e6:.0f} Mvox/s output")
    ident = ChunkedPredictor(lambda x: x.float(), (128, 128, 128), (16, 16, 16), batch_size=8)
    ident.predict(vol)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    back = ident.predict(vol)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"gather + scatter only: {dt * 1e3:.1f} ms ({(216 * 128 ** 3 * 8 + 512 ** 3 * 4) / dt / 1e9:.0f} GB/s algorithmic), identity round trip exact: {torch.equal(back, vol)}")


def bench_prepost(reps=5):
    """HBM-bound scans of biapy_amd.prepost on a 512^3 float32 volume (537 MB)."""
    from biapy_amd import prepost

    n = 512 ** 3
    x = torch.rand(n, device=DEV) * 100
    st = L.stream_ptr()
    ws = torch.empty(int(lib.bpx_select_workspace()), dtype=torch.uint8, device=DEV)
    out = torch.empty(1, device=DEV)
    ms = timeit(lambda: L.check(lib.bpx_select_kth_f32(x.data_ptr(), n, n // 100, out.data_ptr(), ws.data_ptr(), st)), reps)
    print(f"select_kth (4 passes) 512^3: {ms * 1e3:9.1f} us  {4 * n * 4 / ms / 1e6:8.1f} GB/s")
    edges = torch.linspace(0, 100, 257, device=DEV)
    counts = torch.zeros(256, dtype=torch.int64, device=DEV)
    ms = timeit(lambda: L.check(lib.bpx_histogram_f32(x.data_ptr(), n, 0.0, 100.0, 256, edges.data_ptr(), counts.data_ptr(), st)), reps)
    print(f"histogram256          512^3: {ms * 1e3:9.1f} us  {n * 4 / ms / 1e6:8.1f} GB/s")
    o8 = torch.empty(n, dtype=torch.uint8, device=DEV)
    ms = timeit(lambda: L.check(lib.bpx_threshold_u8(x.data_ptr(), n, 50.0, o8.data_ptr(), st)), reps)
    print(f"threshold -> u8       512^3: {ms * 1e3:9.1f} us  {n * 5 / ms / 1e6:8.1f} GB/s")
    o = torch.empty_like(x)
    ms = timeit(lambda: L.check(lib.bpx_clip_affine_f32(x.data_ptr(), n, 1.0, 99.0, 50.0, 20.0, o.data_ptr(), st)), reps)
    print(f"clip + normalise      512^3: {ms * 1e3:9.1f} us  {n * 8 / ms / 1e6:8.1f} GB/s")
    part = torch.empty(lib.bpx_scan_blocks(n), dtype=torch.float64, device=DEV)
    ms = timeit(lambda: L.check(lib.bpx_moment_f32(x.data_ptr(), n, 50.0, 2, part.data_ptr(), st)), reps)
    print(f"moment (double acc.)  512^3: {ms * 1e3:9.1f} us  {n * 4 / ms / 1e6:8.1f} GB/s")
    del prepost


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rcan":
        bench_rcan()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rcan_x4":
        bench_rcan_x4_train()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rcan_fwd":
        bench_rcan_fwd_only()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rcan_train":
        bench_rcan_train_only()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "convt":
        bench_convt()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "chunked":
        bench_chunked()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "prepost":
        bench_prepost()
        sys.exit(0)
    main()
