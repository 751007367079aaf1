"""Idle time between consecutive kernels of one training step (rocprofv3 --kernel-trace --output-format csv):
prints the largest gaps and the totals, to find host- or dependency-bound stretches of the step."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = adam[-3], adam[-2]                      # one full step between two optimizer launches
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
gaps = []
for p, q in zip(step[:-1], step[1:]):
    g = int(q["Start_Timestamp"]) - int(p["End_Timestamp"])
    gaps.append((g, p["Kernel_Name"][:50], q["Kernel_Name"][:50]))
print(f"step: {len(step)} launches, wall {(t1 - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
print(f"gaps: mean {sum(g for g, _, _ in gaps) / len(gaps) / 1e3:.2f} us; > 5 us: {sum(1 for g, _, _ in gaps if g > 5000)}")
for g, p, q in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:7.1f} us  after {p}  before {q}")

# mean gap by the kernel that precedes it (which kernels leave the chip idle behind them)
import re
by = {}
for g, p, q in gaps:
    k = re.sub(r"^_Z\d+|I[Dt].*|<.*|\(.*", "", p)
    d = by.setdefault(k, [0, 0])
    d[0] += g; d[1] += 1
print("mean gap BEHIND a kernel (us), count, total (us):")
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {k[:44]:44s} {t / n / 1e3:6.2f} {n:4d} {t / 1e3:8.1f}")

bq, pairs = {}, {}
for g, p, q in gaps:
    kp = re.sub(r"^_Z\d+|I[Dt].*|<.*|\(.*", "", p)[:28]
    kq = re.sub(r"^_Z\d+|I[Dt].*|<.*|\(.*", "", q)[:28]
    d = bq.setdefault(kq, [0, 0]); d[0] += g; d[1] += 1
    d = pairs.setdefault((kp, kq), [0, 0]); d[0] += g; d[1] += 1
print("mean gap IN FRONT OF a kernel (us), count, total (us):")
for k, (t, n) in sorted(bq.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {k:44s} {t / n / 1e3:6.2f} {n:4d} {t / 1e3:8.1f}")
print("by (previous, next) pair:")
for (kp, kq), (t, n) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {kp:28s} -> {kq:28s} {t / n / 1e3:6.2f} {n:4d} {t / 1e3:8.1f}")
