#!/bin/bash
# Per-kernel time of the default bench step in fp16 (two-term weights + loss scaling) and in bf16 mode, one box:
# where the fp16 mode's extra time goes.  Usage (through gpurun): bash tools/dtype_kernel_diff.sh
export TMPDIR=/tmp W2V2_BENCH_NO_FAMILY_PASS=1
ROOT=$PWD
cd /tmp
for dt in f16 bf16; do
  rm -rf /tmp/dkd_$dt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dkd_$dt -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --dtype $dt --steps 20 --warmup 4 > /tmp/dkd_$dt.log 2>&1
done
python3 - <<'PY'
import csv, glob, re
def load(dt):
    f = glob.glob(f"/tmp/dkd_{dt}/**/r_kernel_stats.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r"DF16_|DF16b|<[^>]*>|I[tf]L?[^E]*E", "", r["Name"])[:48]
        name = re.sub(r"^_Z\d+", "", name)
        out[name] = out.get(name, 0.0) + float(r["TotalDurationNs"]) / 24e6
    return out
a, b = load("f16"), load("bf16")
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, 0) + b.get(k, 0)))
print(f"{'kernel':50s} {'f16 ms/step':>12s} {'bf16 ms/step':>12s} {'diff':>8s}")
for k in keys[:28]:
    print(f"{k:50s} {a.get(k, 0):12.3f} {b.get(k, 0):12.3f} {a.get(k, 0) - b.get(k, 0):8.3f}")
print(f"{'total':50s} {sum(a.values()):12.3f} {sum(b.values()):12.3f} {sum(a.values()) - sum(b.values()):8.3f}")
PY
