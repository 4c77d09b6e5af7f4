"""Round 6, first GPU contact of the role-split fused backward kernel: parity against the separate kernels, timing against the serial fused kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_checks as K

bad = 0
for args in [dict(mix=True, B=1, S=(34, 38, 44), Ct=48, planar=True, rs=1), dict(mix=True, B=2, S=(32, 32, 32), Ct=48, planar=True, rs=1),
             dict(mix=False, B=2, S=(32, 32, 48), Ct=48, planar=False, rs=1), dict(mix=True, B=2, S=(32, 32, 32), Ct=48, planar=True, act=2, rs=1),
             dict(mix=True, B=2, S=(32, 32, 32), Ct=16, rs=2), dict(mix=False, B=1, S=(36, 34, 40), Ct=16, rs=2), dict(mix=True, B=3, S=(32, 32, 32), Ct=16, act=3, rs=2),
             dict(mix=True, B=4, S=(64, 64, 64), Ct=48, planar=True, rs=1), dict(mix=True, B=4, S=(64, 64, 64), Ct=16, rs=2)]:
    for r in K.check_bwd_fused(**args):
        ok = r["err"] <= r["tol"] if isinstance(r, dict) else True
        print(r, flush=True)
        bad += 0 if ok else 1
print("FAILED" if bad else "ALL OK", bad)
