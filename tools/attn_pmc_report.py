#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one directory per pass) per kernel: mean counter value per launch."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
disp = defaultdict(lambda: defaultdict(set))          # kernel -> counter -> dispatch ids (launches are COUNTED, not assumed)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "attn" not in name:
                continue
            short = name[name.index("attn"):][:22]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            disp[short][r["Counter_Name"]].add((f, r.get("Dispatch_Id", r.get("Correlation_Id", len(acc[short][r["Counter_Name"]])))))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        # rocprofv3 emits one row per (dispatch, counter[, dimension]): sum rows of a dispatch = total; here mean per row x rows/dispatch
        n = max(1, len(disp[k][c]))
        print(f"   {c:28s} rows {len(v):5d}  launches {n:4d}  sum/launch {sum(v) / n:16.0f}")
