#!/usr/bin/env python3
"""Every launch of ONE training step in stream order (rocprofv3 --kernel-trace --output-format csv): index, start offset,
duration, idle gap in front of it, kernel.  The map for launch merging / fusion work (which small kernels sit where).
    rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -- python3 bench.py --no-cpu-baseline --no-also --no-eer --steps 6 --warmup 3
    python tools/step_sequence.py /tmp/seq [step_from_the_end=2]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = adam[-back - 1], adam[-back]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
prev_end = int(rows[a]["End_Timestamp"])


def short(n):
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_Z(\d+)", n)
    if m:
        k = int(m.group(1))
        s = m.end()
        n = n[s:s + k] + ("<f16>" if "DF16_" in n[s + k:s + k + 12] else "")
    return re.sub(r"\(.*", "", n)[:46]


tot_k = tot_g = 0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    prev_end = e
    tot_k += e - s
    tot_g += gap
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:6.1f}  {short(r['Kernel_Name'])}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
print(f"# {len(step)} launches; kernels {tot_k / 1e3:.1f} us; gaps {tot_g / 1e3:.1f} us (traced)")
