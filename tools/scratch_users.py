"""Kernels of a traced run that use scratch (private segment): each such dispatch costs ~8 us extra on this stack
(tools/probes/scratch_probe.hip).  Usage: python tools/scratch_users.py <rocprofv3 csv dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = collections.Counter(); sz = {}
for r in csv.DictReader(open(f)):
    s = int(r.get("Scratch_Size") or r.get("Private_Segment_Size") or 0)
    if s > 0:
        n[r["Kernel_Name"][:90]] += 1
        sz[r["Kernel_Name"][:90]] = s
for k, c in n.most_common():
    print(f"{c:6d} launches  scratch {sz[k]:5d} B  {k}")
print("columns:", [c for c in csv.DictReader(open(f)).fieldnames if "cratch" in c or "rivate" in c])
