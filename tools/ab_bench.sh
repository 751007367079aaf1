#!/bin/bash
# A/B of two builds of libw2v2hip.so inside ONE gpurun call (boxes differ by +-4 %): alternates
#   A = tools/ab/lib_ab_old.so (a copy of an earlier build)   B = the in-tree library
# Usage: bash tools/ab_bench.sh [rounds] [bench.py args...]
R=${1:-3}; shift
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then export W2V2_LIB_AB=$PWD/tools/ab/lib_ab_old.so; else unset W2V2_LIB_AB; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
  done
done
