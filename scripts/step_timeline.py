"""Timeline of ONE graph-replayed training step from a rocprofv3 kernel trace (profiling helper, not product code).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --mode train --feed device --steps 6 --warmup 3 ...
    python scripts/step_timeline.py $(find out -name "t_kernel_trace.csv") > profiles/rNN_step_timeline.txt

Takes the LAST run of dispatches between two AdamW launches (= the last timed step), and prints every dispatch with its start offset, duration,
the idle gap in front of it and its grid; then the totals (kernel time, gaps, launches).
"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", name)
    if m:
        return m.group(1) + "<" + name[m.end():][:60] + ">"
    return name.split("(")[0][:90]


def main(path: str) -> None:
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    ends = [i for i, n in enumerate(names) if "adam_step_inc" in n]
    if len(ends) < 2:
        raise SystemExit("fewer than two optimizer steps in the trace")
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    prev_end, busy, gaps = t0, 0, 0
    print(f"# one graph-replayed train step: {len(step)} dispatches (between the last two adam_step_inc launches of {path.split('/')[-1]})")
    print("#  start us   dur us   gap us  grid x block  kernel")
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = s - prev_end
        busy += e - s
        gaps += max(gap, 0)
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f}  {grid:>9} x {wg:<5} {short(r['Kernel_Name'])}")
        prev_end = max(prev_end, e)
    print(f"# step span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle gaps {gaps / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
