#!/bin/bash
# Every A/B switch of the library against the default, interleaved on one box (stale tunings show up here: the driver /
# firmware under the kernels changes between rounds).  bench.py --steps 30; two passes.
SW="NONE W2V2_G3_NONPERSISTENT=1 W2V2_NO_DEFER=1 W2V2_EPI_WT=0 W2V2_PH_LATE=1 W2V2_NO_LN_FOLD=1 W2V2_LN_NO_QUAD=1 W2V2_NO_LN_PREFETCH=1 W2V2_NO_WGRAD_PAIRS=1 W2V2_ATTN_NO_XCD_REMAP=1 W2V2_CONV0_NO_GRAM=1 W2V2_ADAM_U=4 W2V2_ADAM_U=1 W2V2_G3N=1024 W2V2_G3N=256 W2V2_NO_GEMM_PH=1 W2V2_NO_WGRAD_PH=1 W2V2_NO_POSCONV_DIRECT=1 W2V2_ATTN_GEOM=64 W2V2_ATTN_KV_NO_DMA=1"
for pass in 1 2; do for s in $SW; do
  if [ $s = NONE ]; then E=""; else E="$s"; fi
  env $E python bench.py --no-cpu-baseline --no-also --no-eer --no-families --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$s', d['ms_per_step'], d['value'])"
done; done
