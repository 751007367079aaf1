#!/bin/bash
# Whole-tree A/B inside ONE gpurun call (boxes differ by +-4 %): A = tools/ab/tree (an export of an earlier commit with its
# own built library: `git archive <commit> | tar -x -C tools/ab/tree` + the library built by tools/build_ab_old.sh),
# B = the working tree; alternated R times.   Usage: bash tools/ab_tree.sh [rounds] [bench.py args...]
R=${1:-3}; shift
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then B=tools/ab/tree/bench.py; else B=bench.py; fi
    python3 $B --no-cpu-baseline --no-also $(grep -q -- "--no-eer" $B && echo --no-eer) "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('ms_per_step_median'), d['value'])"
  done
done
