#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for m in -1 0 6 14 20; do for v in 0 1; do
  if [ $v = 1 ]; then export W2V2_NO_WGRAD_STREAMK=1; else unset W2V2_NO_WGRAD_STREAMK; fi
  if [ $m = -1 ]; then unset W2V2_WGRAD_SK_MARGIN; else export W2V2_WGRAD_SK_MARGIN=$m; fi
  python3 bench.py --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('margin $m streamK off=$v', d['ms_per_step'], d['value'])"
done; done
