#!/bin/bash
# configs[3] per-GPU share: A = tools/ab/lib_ab_old.so (a copy of an earlier build), B = the in-tree library, ABAB on one box
for i in 1 2 3; do for v in A B; do
  if [ $v = A ]; then export W2V2_LIB_AB=$PWD/tools/ab/lib_ab_old.so; else unset W2V2_LIB_AB; fi
  python bench.py --model large --seconds 5 --batch 32 --no-cpu-baseline --no-also --no-eer --no-families --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
done; done
