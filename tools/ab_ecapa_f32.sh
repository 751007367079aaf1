#!/bin/bash
# ECAPA-TDNN step (bench.py --model ecapa, f32): register-staged f32 GEMM (W2V2_F32_NO_DMA=1) against the library's
# choice, ABAB on one box.   gpurun -- 'bash tools/ab_ecapa_f32.sh'
mkdir -p gpurun_out
out=gpurun_out/ab_ecapa_f32.txt
: > $out
for r in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export W2V2_F32_NO_DMA=1; else unset W2V2_F32_NO_DMA; fi
    line=$(python3 bench.py --model ecapa --steps 20 --warmup 5 2>/dev/null | tail -1)
    echo "$v $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['value'])" "$line")" >> $out
  done
done
unset W2V2_F32_NO_DMA
cat $out
