"""HBM bytes per launch of the conv entry points from two rocprofv3 --pmc passes over `python bench.py ...`.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python scripts/pmc_traffic.py gpurun_out/pmc_f/*/p_results.db gpurun_out/pmc_w/*/p_results.db > profiles/pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide (16 B/lane) coalesced reads at 64 B, so it is doubled for the conv kernels, whose global reads are
all 16 B/lane.  WRITE_SIZE is used as reported.  A wgrad call is its MFMA kernel plus the reduce kernel (and both k=3 and k=1 launches): pass the number of C-ABI calls of
the profiled run as a JSON object in argv[3] to get bytes per CALL, the unit bench.py's roofline uses.
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict

# kernel name -> C-ABI entry point.  conv3_kernel<T,TZ,TY,TX,NS,EPI,ACTK> and conv3_lp_kernel<TZ,TY,TX,NS,EPI,ACTK>: EPI 0 = fwd, 1 = dgrad
def _conv_rx(epi):
    """conv3_lp_kernel<TZ, TY, TX, NS, EPI, ACTK, F16, TF16> and conv3_kernel<T, TZ, TY, TX, NS, EPI, ACTK, TT>, demangled or - rocprofv3 leaves the
    names with _Float16 arguments mangled - as conv3_kernelIDF16_Li4ELi4ELi8ELi4ELi<EPI>E... / conv3_lp_kernelILi4E...Li<EPI>E..."""
    e = str(epi)
    zm = r"|conv3_z[ms]_kernel(<|I)" if epi == 0 else ""      # the z-marching forward kernel of the level-0 layers (round 5): forward only
    return re.compile(r"conv3_lp_kernel<\d+, \d+, \d+, \d+, " + e + r",|conv3_kernel<[^,>]+, \d+, \d+, \d+, \d+, " + e + r"," + zm +
                      r"|conv3_lp_kernelI(?:Li\d+E){4}Li" + e + r"E|conv3_kernelI(?:DF16_|t|f)(?:Li\d+E){4}Li" + e + r"E")


GROUPS = [
    ("bpx_conv3d_fwd", _conv_rx(0)),
    ("bpx_conv3d_dgrad", _conv_rx(1)),
    ("bpx_conv3d_wgrad", re.compile(r"wgrad_(sdm?_)?kernel(<|I)|wgrad_reduce(_batch)?_kernel")),
    # the fused backward (dgrad + wgrad of one conv in one pass, round 4); its partial slabs are summed by the same batched reduce launch as the
    # wgrad calls' (counted with them above)
    ("bpx_conv3d_bwd_fused", re.compile(r"conv3_bwd(_rs)?_kernel(<|I)")),   # serial and role-split (round 6) forms
    # the sliding-window blend / gather (tests/bench_kernels.py merge: 512 x 128^3 <-> 512^3; 16 B/lane row kernels, same doubling)
    ("bpx_merge3d_blend", re.compile(r"merge3d_row_kernel<")),
    ("bpx_crop3d_gather", re.compile(r"crop3d_row_kernel<")),
]


CALLS = defaultdict(int)


def collect(db, counter):
    con = sqlite3.connect(db)
    acc = defaultdict(list)
    for name, val in con.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        for g, rx in GROUPS:
            if rx.search(name):
                acc[g].append(val)
                if "reduce" not in name:
                    CALLS[(db, g)] += 1          # a C-ABI call = one MFMA kernel (+ its reduce kernel for wgrad)
                break
    return acc


def main():
    fdb, wdb = sys.argv[1], sys.argv[2]
    f = collect(fdb, "FETCH_SIZE")
    w = collect(wdb, "WRITE_SIZE")
    out = {}
    launches = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}   # C-ABI launches in the profiled run (a call = >= 1 kernels)
    for g, _ in GROUPS:
        if g not in f:
            continue
        nf = launches.get(g, CALLS[(fdb, g)] or len(f[g]))
        nw = launches.get(g, CALLS[(wdb, g)] or len(w.get(g, [])))
        fetch_raw = 1024.0 * sum(f[g]) / nf
        write = 1024.0 * sum(w[g]) / nw if nw else None
        out[g] = dict(launches=nf, fetch_raw_bytes=round(fetch_raw), fetch_bytes=round(2 * fetch_raw), write_bytes=round(write) if write else None,
                      total_bytes=round(2 * fetch_raw + (write or 0)))
    try:   # stamp the kernel sources the counters belong to (bench.py refuses to quote a file whose stamp differs from the tree's)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from biapy_amd._lib import source_digest
        out["_meta"] = dict(csrc_sha256=source_digest(), tiling_sha256=source_digest(only=("tiling.hip", "bpx_common.h")))
    except Exception as e:  # noqa: BLE001
        out["_meta"] = dict(csrc_sha256=None, error=str(e))
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
