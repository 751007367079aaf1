#!/bin/bash
# configs[3] per-GPU share (wav2vec2-large, 5 s, 32 utterances): weight-gradient groups of 2 vs 4, alternating on one box
for i in 1 2 3; do for g in 2 4; do
  W2V2_WGRAD_GROUP=$g python bench.py --model large --seconds 5 --batch 32 --no-cpu-baseline --no-also --no-eer --no-families --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group $g', d['ms_per_step'], d['value'], d['config']['final_loss'])"
done; done
