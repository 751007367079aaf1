#!/bin/bash
# Per-kernel time table of any bench.py configuration under rocprofv3 (one box, through gpurun):
#   bash tools/kernel_table.sh <timed steps> <warm-up> [bench.py args...]     e.g. 10 3 --model large --seconds 5 --batch 32
K=${1:-10}; W=${2:-3}; shift; shift
export TMPDIR=/tmp W2V2_BENCH_NO_FAMILY_PASS=1
R=$PWD; cd /tmp; rm -rf /tmp/ktab
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktab -o r -- python3 $R/bench.py --no-cpu-baseline --no-also --steps $K --warmup $W "$@" > /tmp/ktab.log 2>&1
python3 - $((K + W)) <<'PY'
import csv, glob, sys
n = float(sys.argv[1])
rows = list(csv.DictReader(open(glob.glob("/tmp/ktab/**/r_kernel_stats.csv", recursive=True)[0])))
print("kernel time per step (all steps incl. warm-up): %.3f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e6))
print(" ms/step  calls/step    avg_us  kernel")
for r in rows[:26]:
    print("%8.3f %11.1f %9.1f  %s" % (float(r["TotalDurationNs"]) / n / 1e6, int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, r["Name"][:80]))
PY
