#!/usr/bin/env python3
"""Fold gpurun_out/f32_dma_sweep.txt (tools/f32_dma_sweep.sh) into one table: a row per product of the ECAPA step, a column
per forced kernel / tile, us per launch; last line = the step's total per column and the best-per-shape total."""
import sys
from collections import OrderedDict
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/f32_dma_sweep.txt"
cols, data, cur = [], OrderedDict(), None
for line in open(path):
    if line.startswith("#####"):
        cur = line[5:].strip().split(" ")[0].replace("F32_TILE=", "t").replace("W2V2_F32_NO_DMA=1", "old")
        cols.append(cur)
        continue
    f = line.split()
    if len(f) >= 13 and f[0].isdigit():
        key = tuple(int(x) for x in f[:9])
        data.setdefault(key, {})[cur] = (float(f[10]), int(f[9]))
print(f"{'M':>6} {'N':>5} {'K':>6} {'b':>3} tA tB {'sp':>3} ep  n " + " ".join(f"{c:>8}" for c in cols) + "   best")
tot = {c: 0.0 for c in cols}
best_tot = 0.0
for key, d in sorted(data.items(), key=lambda kv: -max(v[0] * v[1] for v in kv[1].values())):
    n = next(iter(d.values()))[1]
    b = min(d, key=lambda c: d[c][0])
    print(f"{key[0]:6d} {key[1]:5d} {key[2]:6d} {key[3]:3d} {key[4]:2d} {key[5]:2d} {key[6]:3d} {key[7]:2d} {n:2d} " +
          " ".join(f"{d[c][0]:8.1f}" if c in d else " " * 8 for c in cols) + f"   {b}")
    for c in cols:
        if c in d:
            tot[c] += d[c][0] * n
    best_tot += d[b][0] * n
print(" " * 37 + " ".join(f"{tot[c] / 1e3:8.2f}" for c in cols) + f"   ms/step; best per shape {best_tot / 1e3:.2f}")
