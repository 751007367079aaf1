#!/usr/bin/env python3
"""HBM traffic per launch of every kernel from two rocprofv3 PMC passes (rocpd sqlite results):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dirF> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dirW> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py <dirF> <dirW> > profiles/rNN_pmc_hbm_traffic.json
gfx950 correction (MI355X_MICROARCH.md, HBM section; calibrated on adam_kernel whose traffic is known exactly):
FETCH_SIZE counts 64 B per 128-B request -> read bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE (KiB) is exact."""
import glob
import json
import re
import sqlite3
import sys


def per_kernel(d, counter):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    q = ("select k.name, count(*), sum(c.value) from counters_collection c join kernels k "
         "on k.dispatch_id = c.dispatch_id where c.counter_name = ? group by k.name")
    return {re.sub(r"\(.*", "", n): (cnt, tot) for n, cnt, tot in cur.execute(q, (counter,))}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": __doc__.strip().split("\n\n")[0].split("\n", 1)[0] + " Read bytes = 2 * FETCH_SIZE KiB (gfx950), "
                   "write bytes = WRITE_SIZE KiB; separate --pmc passes of bench.py --steps 2 --warmup 1 (B=66).",
           "kernels": {}}
    for name in sorted(f, key=lambda n: -(2 * f[n][1] + w.get(n, (0, 0))[1])):
        n, fk = f[name]
        wk = w.get(name, (n, 0.0))[1]
        out["kernels"][name.replace("void ", "")] = {
            "launches": n, "fetch_kib_raw": round(fk / n, 1), "write_kib": round(wk / n, 1),
            "hbm_bytes_per_launch": int((2 * fk + wk) / n * 1024)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
