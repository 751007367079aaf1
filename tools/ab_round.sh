#!/bin/bash
# Same-box A/B of the whole tree against the round-2 tree (git worktree build_ab/${AB_TREE:-r03} with its own built library):
#   A = python3 build_ab/${AB_TREE:-r03}/bench.py     B = python3 bench.py        alternated R times inside ONE gpurun call
R=${1:-3}; shift
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then B=build_ab/${AB_TREE:-r03}/bench.py; else B=bench.py; fi
    python3 $B --no-cpu-baseline $([ $v = B ] && echo --no-also) "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
  done
done
