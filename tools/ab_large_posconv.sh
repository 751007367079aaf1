#!/bin/bash
# configs[3] per-GPU share: positional convolution through the implicit GEMM (W2V2_NO_POSCONV_DIRECT=1) vs the image-resident kernel
for i in 1 2 3; do for g in 1 0; do
  if [ $g = 1 ]; then export W2V2_NO_POSCONV_DIRECT=1; else unset W2V2_NO_POSCONV_DIRECT; fi
  python bench.py --model large --seconds 5 --batch 32 --no-cpu-baseline --no-also --no-eer --no-families --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('implicit_gemm=$g', d['ms_per_step'], d['value'], d['config']['final_loss'])"
done; done
