#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite .db written by `rocprofv3 --kernel-trace --stats`) as a
small text table: per-kernel calls / total / average / share, plus the dominant kernel's launch
geometry.  Usage: tools/rocprof_summary.py <results.db> [out.md] [--title "..."]"""
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    title = "rocprofv3 kernel-trace summary"
    if "--title" in sys.argv:
        title = sys.argv[sys.argv.index("--title") + 1]
        args = [a for a in args if a != title]
    db = args[0]
    out = open(args[1], "w") if len(args) > 1 else sys.stdout
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    print(f"# {title}\n", file=out)
    print(f"source: `{db.split('/')[-1]}` (rocprofv3 --kernel-trace --stats; durations in microseconds)\n", file=out)
    print("| kernel | calls | total us | avg us | % |", file=out)
    print("|---|---:|---:|---:|---:|", file=out)
    for name, calls, tot, avg, pct in rows:
        print(f"| `{name.split('(')[0]}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |", file=out)
    try:
        q = ("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, sgpr_count, count(*), "
             "avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0 from kernels "
             "group by name, grid_x order by sum(duration) desc limit 8")
        print("\n| kernel | grid | wg | lds | scratch | vgpr | sgpr | n | avg us | min us | max us |", file=out)
        print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|", file=out)
        for r in c.execute(q):
            print(f"| `{r[0].split('(')[0]}` | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} | "
                  f"{r[8]:.2f} | {r[9]:.2f} | {r[10]:.2f} |", file=out)
    except sqlite3.Error as e:  # schema differences between rocprofv3 versions
        print(f"\n(per-dispatch table unavailable: {e})", file=out)


if __name__ == "__main__":
    main()
