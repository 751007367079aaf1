#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -k "grouped_weight or wgrad" > $OUT/r03_t12.log 2>&1; tail -8 $OUT/r03_t12.log
for i in 1 2 3; do for v in 1 0; do if [ $v = 1 ]; then export W2V2_NO_WGRAD_STREAMK=1; else unset W2V2_NO_WGRAD_STREAMK; fi; python3 bench.py --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streamK off=$v', d['ms_per_step'], d['value'])"; done; done
unset W2V2_NO_WGRAD_STREAMK
timeout 1200 python3 -m pytest tests/test_parity_gpu.py -q -m gpu -k "b66 or base_16bit or tiny_all" > $OUT/r03_t12b.log 2>&1; tail -4 $OUT/r03_t12b.log
