#!/usr/bin/env python3
"""Static instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing, per basic block.

    python scripts/isa_blocks.py /tmp/lean.s <kernel symbol substring> [--ops]

Prints every basic block (label, instruction counts by class, the branch that ends it); with --ops the mnemonic histogram of the
whole kernel.  The counts are static: multiply by the trip counts read off the branch structure."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0])
    blocks, cur, name = [], collections.Counter(), "entry"
    ops = collections.Counter()
    last = ""
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
            blocks.append((name, cur, last))
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur, last))
            cur, name, last = collections.Counter(), m.group(1), ""
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        cur[classify(op)] += 1
        ops[op] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            last = s.split(";")[0]
    tot = collections.Counter()
    for name, c, last in blocks:
        n = sum(c.values())
        tot.update(c)
        if n >= 12:
            print("%-12s %5d  %s   -> %s" % (name, n, " ".join("%s=%d" % kv for kv in sorted(c.items())), last))
    print("TOTAL", sum(tot.values()), dict(tot))
    if "--ops" in sys.argv:
        for op, n in ops.most_common(60):
            print("  %-28s %d" % (op, n))


main()
