timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import kernel_checks as K
for a in [(True, 2, (32, 32, 32), 48, 16, True), (True, 2, (32, 32, 32), 48, 16, False), (False, 2, (32, 32, 32), 48, 16, True), (False, 1, (34, 38, 52), 48, 16, False), (True, 1, (40, 40, 44), 96, 32, True), (True, 2, (32, 32, 36), 16, 32, False), (False, 1, (64, 64, 64), 96, 32, False)]:
    for r in K.check_wgrad_k1_stream(*a):
        print("ok  " if r["ok"] else "FAIL", r["name"], "%.3e" % r["err"], r["tol"])
PY
