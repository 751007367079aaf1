#!/bin/bash
# Adam grid cap in the step: the round-3 cap (8192 blocks) against the default (no cap), alternating on one box; base and large
for i in 1 2 3 4; do for nb in 8192 0; do
  if [ $nb = 0 ]; then unset W2V2_ADAM_BLOCKS; else export W2V2_ADAM_BLOCKS=$nb; fi
  python bench.py --no-cpu-baseline --no-also --no-eer --no-families --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base  cap $nb', d['ms_per_step'], d['value'])"
done; done
for i in 1 2 3; do for nb in 8192 0; do
  if [ $nb = 0 ]; then unset W2V2_ADAM_BLOCKS; else export W2V2_ADAM_BLOCKS=$nb; fi
  python bench.py --model large --seconds 5 --batch 32 --no-cpu-baseline --no-also --no-eer --no-families --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('large cap $nb', d['ms_per_step'], d['value'])"
done; done
