#!/bin/bash
# Kernel-level A/B inside ONE gpurun call: average duration of the kernels matching <pattern> in the default bench
# step, A = tools/ab/lib_ab_old.so, B = the in-tree library (alternating, <rounds> times).
# Usage: bash tools/ab_prof.sh <pattern> [rounds] [bench.py args...]
PAT=$1; R=${2:-2}; shift; shift
export TMPDIR=/tmp W2V2_BENCH_NO_FAMILY_PASS=1
ROOT=$PWD
cd /tmp
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then export W2V2_LIB_AB=$ROOT/tools/ab/lib_ab_old.so; else unset W2V2_LIB_AB; fi
    rm -rf /tmp/abp
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 20 --warmup 4 "$@" > /tmp/abp.log 2>&1
    F=$(find /tmp/abp -name "r_kernel_stats.csv" | head -1)
    echo "== $v"
    python3 - "$F" "$PAT" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"]) / 1e3:9.2f}')
PY
  done
done
