#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db or *_kernel_trace.csv) into the per-kernel
table committed under profiles/ (name, launches, total ms, avg us, share)."""
import csv
import re
import sqlite3
import sys


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by 3 desc"))


def rows_from_csv(path):
    agg = {}
    for r in csv.DictReader(open(path)):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    return sorted([(k, v[0], v[1], v[1] / v[0], v[2], v[3]) for k, v in agg.items()], key=lambda x: -x[2])


def main():
    path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    tot = sum(r[2] for r in rows)
    print(f"# source: {path}   total kernel time {tot / 1e6:.3f} ms over {steps} step(s) = {tot / 1e6 / steps:.3f} ms/step")
    print(f"{'share':>7} {'ms/step':>9} {'calls':>6} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  kernel")
    for n, c, s, a, mn, mx in rows:
        n = re.sub(r"\(.*", "", n)
        print(f"{100 * s / tot:6.2f}% {s / 1e6 / steps:9.3f} {c:6d} {a / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f}  {n}")


if __name__ == "__main__":
    main()
