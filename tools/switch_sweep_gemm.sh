#!/bin/bash
# The GEMM-side A/B switches against the default, two interleaved passes on one box (re-run after the compiler-wait fix).
SW="NONE W2V2_G3_NONPERSISTENT=1 W2V2_NO_DEFER=1 W2V2_EPI_WT=0 W2V2_PH_LATE=1 W2V2_G3N=1024 W2V2_G3N=256 W2V2_NO_GEMM_PH=1 NONE"
for pass in 1 2; do for s in $SW; do
  if [ $s = NONE ]; then E=""; else E="$s"; fi
  env $E python bench.py --no-cpu-baseline --no-also --no-eer --no-families --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$s', d['ms_per_step'], d['value'])"
done; done
