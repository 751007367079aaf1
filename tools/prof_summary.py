#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db or *_kernel_trace.csv) into the per-kernel
table committed under profiles/ (name, launches, total ms, avg us, share).

    prof_summary.py <db|csv> <steps>                 every launch of the process, divided by <steps>
    prof_summary.py <db> <steps> --steady adam_kernel
        only the last <steps> STEADY-STATE steps: the window from the end of the (steps+1)-th last launch of the named
        once-per-step kernel to the end of its last launch -- start-up work (parameter initialisation copies, buffer
        fills, the one-time weight packing) is left out, so every `calls` entry is a per-step count times <steps>."""
import csv
import re
import sqlite3
import sys


def rows_from_db(path, steady=None, steps=1):
    cur = sqlite3.connect(path).cursor()
    where, note = "", ""
    if steady:
        ends = [r[0] for r in cur.execute("select end from kernels where name like ? order by end", (f"%{steady}%",))]
        if len(ends) > steps:
            where = f" where start > {ends[-steps - 1]} and end <= {ends[-1]}"
            note = f"steady-state window: the last {steps} steps (delimited by {steady})"
        else:
            note = f"only {len(ends)} launches of {steady}: whole process"
    return list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            f"from kernels{where} group by name order by 3 desc")), note


def rows_from_csv(path):
    agg = {}
    for r in csv.DictReader(open(path)):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    return sorted([(k, v[0], v[1], v[1] / v[0], v[2], v[3]) for k, v in agg.items()], key=lambda x: -x[2])


def main():
    path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
    steady = sys.argv[sys.argv.index("--steady") + 1] if "--steady" in sys.argv else None
    rows, note = rows_from_db(path, steady, steps) if path.endswith(".db") else (rows_from_csv(path), "")
    tot = sum(r[2] for r in rows)
    print(f"# source: {path}   total kernel time {tot / 1e6:.3f} ms over {steps} step(s) = {tot / 1e6 / steps:.3f} ms/step")
    if note:
        print(f"# {note}")
    print(f"{'share':>7} {'ms/step':>9} {'calls':>6} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  kernel")
    for n, c, s, a, mn, mx in rows:
        n = re.sub(r"\(.*", "", n)
        print(f"{100 * s / tot:6.2f}% {s / 1e6 / steps:9.3f} {c:6d} {a / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f}  {n}")


if __name__ == "__main__":
    main()
