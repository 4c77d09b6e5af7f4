"""Summarise rocprofv3 sqlite outputs: per-kernel counter means (pmc dbs) or duration stats (kernel-trace dbs).

    python scripts/pmc_report.py gpurun_out/pmc_x/p_results.db [kernel-substring]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    f = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    con = sqlite3.connect(f)
    cur = con.cursor()
    tabs = {r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")}
    if "counters_collection" in tabs and cur.execute("select count(*) from counters_collection").fetchone()[0]:
        acc = defaultdict(lambda: defaultdict(list))
        meta = {}
        for name, cname, val, vg, ag, sg, lds, grid, wg in cur.execute(
            "select kernel_name, counter_name, value, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, grid_size, workgroup_size from counters_collection"):
            if sub in name:
                acc[name][cname].append(val)
                meta[name] = (vg, ag, sg, lds, grid, wg)
        for name, cs in acc.items():
            print(name[:150])
            print("   vgpr=%s agpr=%s sgpr=%s lds=%s grid=%s wg=%s dispatches=%d" % (meta[name] + (len(next(iter(cs.values()))),)))
            for c, v in sorted(cs.items()):
                print("   %-32s mean=%.4g" % (c, sum(v) / len(v)))
    if "kernels" in tabs:
        cols = [c[1] for c in cur.execute("pragma table_info('kernels')")]
        if "duration" in cols:
            acc = defaultdict(list)
            for name, d in cur.execute("select name, duration from kernels"):
                if sub in name:
                    acc[name].append(d)
            tot = sum(sum(v) for v in acc.values())
            print("%-100s %8s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
            for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                print("%-100s %8d %12.1f %12.2f %6.1f%%" % (name[:100], len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, 100 * sum(v) / max(tot, 1)))


if __name__ == "__main__":
    main()
