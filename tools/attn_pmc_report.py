#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one directory per pass) per kernel: mean counter value per launch."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "attn" not in name:
                continue
            short = name[name.index("attn"):][:22]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        # rocprofv3 emits one row per (dispatch, counter[, dimension]): sum rows of a dispatch = total; here mean per row x rows/dispatch
        print(f"   {c:28s} rows {len(v):5d}  sum/launch {sum(v) / 6:16.0f}")
