"""Which kernels wait for ALL their outstanding memory operations inside a loop that also stores?

A wave's VMEM operations retire in order through one counter, so an `s_waitcnt vmcnt(0)` inside a persistent loop is a wait for every store issued
so far: the loop runs at [load latency + write latency] per trip.  hipcc emits that wait wherever its wait-count pass loses track - at the joins of
predicated blocks around loads / stores (bounds tests), behind run-time inner loops.  Round 5: the transposed-conv forward of level 0 spent a third
of its time there (profiles/r05_zmarch_ab.txt, section 10).  Usage (no GPU needed):

    python scripts/isa_store_waits.py            # compiles csrc/*.hip to ISA under /tmp and lists loops with stores and vmcnt(0) waits
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"


def demangle(n):
    try:
        return subprocess.run([FILT, n], capture_output=True, text=True).stdout.strip()[:120]
    except OSError:
        return n


def main():
    procs = []
    for src in sorted(glob.glob(os.path.join(ROOT, "biapy_amd", "csrc", "*.hip"))):
        out = "/tmp/isa_" + os.path.basename(src)[:-4] + ".s"
        procs.append((out, subprocess.Popen([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only", src,
                                             "-o", out], stderr=subprocess.DEVNULL)))
    rows = []
    for out, pr in procs:
        pr.wait()
        text = open(out).read()
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
            name, body = m.group(1), m.group(2).split("\n")
            # a loop = its header block + every block the compiler annotated "in Loop: Header=<that block>"
            heads = {}
            cur = None
            for i, l in enumerate(body):
                mm = re.match(r"^\.L(BB\d+_\d+):(.*)", l)
                if mm:
                    lab, rest = mm.group(1), mm.group(2)
                    hm = re.search(r"Header=(BB\d+_\d+)", rest)
                    cur = lab if "Loop Header" in rest else (hm.group(1) if hm else None)
                    if "Loop Header" in rest:
                        heads.setdefault(lab, [])
                if cur is not None and cur in heads:
                    heads[cur].append(l)
            for lab, reg in heads.items():
                st = sum(1 for l in reg if re.search(r"(global|buffer)_store", l))
                ld = sum(1 for l in reg if re.search(r"(global|buffer)_load", l))
                w0 = sum(1 for l in reg if "s_waitcnt vmcnt(0)" in l)
                if st and w0:
                    rows.append((w0, st, ld, len(reg), os.path.basename(out)[4:-2], demangle(name)))
    print("vmcnt(0)  stores  loads  lines  file         kernel   (loops that store AND wait for everything)")
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for r in sorted(rows, reverse=True):
        if only and only not in r[5]:
            continue
        print("%8d %7d %6d %6d  %-12s %s" % r)


if __name__ == "__main__":
    sys.exit(main())
