"""Run every kernel/network parity check and print a table (does not stop at the first failure).

    python tests/gpu_diag.py [--net] [--out gpurun_out/diag.txt]
"""
import argparse
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _emit(rows, out):
    lines = []
    nbad = 0
    for r in rows:
        nbad += 0 if r["ok"] else 1
        lines.append("%-4s %-78s err=%.3e tol=%.1e %s" % ("ok" if r["ok"] else "FAIL", r["name"], r["err"], r["tol"], r.get("extra", "")))
    lines.append(f"{len(rows) - nbad}/{len(rows)} checks passed")
    text = "\n".join(lines)
    print(text)
    if out:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        open(out, "w").write(text + "\n")
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="run one group alone: rcan_up | wide48 | padded | dice")
    a = ap.parse_args()
    import kernel_checks as K

    gt = np.load(os.path.join(ROOT, "tests", "golden", "tiling_golden.npz"))
    gr = np.load(os.path.join(ROOT, "tests", "golden", "resunet_golden.npz"))
    rows = []

    def run(fn, *args, **kw):
        try:
            rows.extend(fn(*args, **kw))
        except Exception as e:  # keep going: a GPU call is expensive
            rows.append(dict(name=f"{fn.__name__}{args[:3]}", err=float("nan"), tol=0, ok=False, extra="EXC " + repr(e)[:300]))
            traceback.print_exc()
        torch.cuda.synchronize()

    RCAN_UP = [(2, torch.bfloat16, 1), (2, torch.float16, 8), (3, torch.bfloat16, 8), (4, torch.float16, 1), (4, torch.float16, 8)]
    if a.only == "rcan_up":                  # the RCAN x-scale stage's training rows alone (added after the round's full table was collected)
        for sc, dtype, grp in RCAN_UP:
            run(K.check_rcan_upscale_train, sc, dtype, grp)
        return _emit(rows, a.out)
    if a.only == "dice":                     # the north-star Dice rows alone (toy net + the benched architecture)
        run(K.check_dice_parity_trained)
        run(K.check_dice_benched_arch)
        return _emit(rows, a.out)
    if a.only == "padded":                   # FEATURE_MAPS [52, 68, 84] (CartoCell template): zero-padded to [64, 80, 96] inside the engine - module vs the CPU oracle on the true widths
        for dtype in (torch.float32, torch.bfloat16, torch.float16):
            run(K.check_network_padded_widths, dtype)
        return _emit(rows, a.out)
    if a.only == "wide48":                   # FEATURE_MAPS [48, 64]: the wide head (48 features), 112-channel concatenation - against the reference fixture
        gv = np.load(os.path.join(ROOT, "tests", "golden", "resunet_variants_golden.npz"))
        for dtype in (torch.float32, torch.bfloat16, torch.float16):
            run(K.check_resunet_variant, dtype, "wide48", gv)
        return _emit(rows, a.out)
    run(K.check_selftest)
    run(K.check_tiling, gt)
    run(K.check_merge_sharded)
    L = K.L
    for dt in (L.F32, L.BF16):
        run(K.check_conv3d_fwd, dt, 2, (8, 8, 16), 16, 16, norm=True, sc_C=0)
        run(K.check_conv3d_fwd, dt, 1, (8, 12, 20), 48, 16, norm=True, sc_C=48, slices=True)
        run(K.check_conv3d_fwd, dt, 2, (6, 8, 8), 32, 64, norm=False, sc_C=1)
        run(K.check_conv3d_fwd, dt, 1, (4, 4, 8), 64, 128, norm=True, sc_C=64)
        run(K.check_conv3d_fwd, dt, 1, (32, 32, 32), 16, 32, norm=True, sc_C=16)
        run(K.check_conv3d_dgrad, dt, 2, (8, 8, 16), 48, 16)
        run(K.check_conv3d_dgrad, dt, 1, (4, 8, 8), 32, 64)
        run(K.check_conv3d_wgrad, dt, 2, (8, 8, 16), 16, 16, k=3, norm=True)
        run(K.check_conv3d_wgrad, dt, 1, (8, 12, 20), 48, 32, k=3, norm=True)
        run(K.check_conv3d_wgrad, dt, 1, (4, 8, 8), 64, 64, k=3, norm=False)
        run(K.check_conv3d_wgrad, dt, 2, (8, 8, 16), 48, 16, k=1, norm=False)
        if dt == L.BF16:
            run(K.check_conv3d_wgrad, dt, 2, (8, 8, 16), 16, 16, k=3, norm=True, use_tr=0)
        run(K.check_conv1x1, dt, 2, 1000, 16, 48, with_coef=True)
        run(K.check_conv1x1, dt, 1, 300, 128, 384, with_coef=False)
        run(K.check_convT, dt, 2, (4, 6, 8), 32)
        run(K.check_convT, dt, 1, (2, 2, 2), 256)
        run(K.check_norm_pool_head, dt)
    run(K.check_bwd_fused, True, 2, (32, 32, 32), 16)
    run(K.check_bwd_fused, True, 1, (34, 38, 44), 48, True)
    run(K.check_bwd_fused, False, 1, (36, 34, 40), 16)
    run(K.check_bwd_fused, True, 1, (32, 32, 32), 32, False, 0, 1, 32)
    run(K.check_wgrad_k1_stream, True, 2, (32, 32, 32), 48, 16, True)
    run(K.check_wgrad_k1_stream, False, 1, (34, 38, 52), 48, 16, False)
    run(K.check_wgrad_k1_stream, True, 1, (40, 40, 44), 96, 32, True)
    run(K.check_pw_stream, True, 2, 131072, 16, True)
    run(K.check_pw_stream, True, 3, 98304, 32, True)
    run(K.check_c1_wgrad_nb, True, 2, (12, 20, 36))
    run(K.check_norm_bwd_finalize_deferred, 4, 768, 16)
    run(K.check_norm_bwd_finalize_deferred, 3, 5000, 48)
    run(K.check_fused_adam)
    run(K.check_f16_saturation)
    run(K.check_convT_wgrad_stream, True, 3, (24, 28, 32), 64)
    run(K.check_convT_wgrad_stream, False, 1, (40, 36, 64), 32)
    if a.net:
        run(K.check_sliding_window, torch.float32)
        run(K.check_sliding_window, torch.bfloat16)
        run(K.check_dice_parity_trained)
        run(K.check_dice_benched_arch)                                     # round 5: the Dice bar on the five-level cfg-2 architecture, 64^3 and the benched 128^3
        for dtype in (torch.float32, torch.bfloat16, torch.float16):      # float16 = the mixed training mode (fp16 forward, bf16 gradients): the benched one
            run(K.check_network, dtype, None, None, None, golden=gr)
            run(K.check_network, dtype, [16, 32, 64, 128, 256], (64, 64, 64), 1, seed=3)
        gd = np.load(os.path.join(ROOT, "tests", "golden", "resunet_dropout_golden.npz"))
        for dtype in (torch.float32, torch.bfloat16, torch.float16):      # MODEL.DROPOUT_VALUES > 0: the reference's training-mode step with its masks made explicit
            run(K.check_network_dropout, dtype, gd)
        for dtype in (torch.float32, torch.bfloat16, torch.float16):      # the benched shape itself: 128^3, logits / loss / gradients (per level in 16 bit)
            run(K.check_network_cfg2_benched_shape, dtype)
        for sc, dtype, grp in RCAN_UP:
            run(K.check_rcan_upscale_train, sc, dtype, grp)
        gv = np.load(os.path.join(ROOT, "tests", "golden", "resunet_variants_golden.npz"))
        for dtype in (torch.float32, torch.bfloat16, torch.float16):      # template widths: GEMM-fed head (fixture), zero-padded widths (oracle on the true widths)
            run(K.check_resunet_variant, dtype, "wide48", gv)
            run(K.check_network_padded_widths, dtype)
    return _emit(rows, a.out)


if __name__ == "__main__":
    sys.exit(main())
