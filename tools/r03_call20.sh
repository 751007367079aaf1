#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attention or attn" > $OUT/r03_t20.log 2>&1; tail -3 $OUT/r03_t20.log
for i in 1 2 3; do for v in 1 0; do if [ $v = 1 ]; then export W2V2_ATTN_NO_XCD_REMAP=1; else unset W2V2_ATTN_NO_XCD_REMAP; fi; python3 bench.py --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn remap off=$v', d['ms_per_step'], d['value'])"; done; done
python3 tools/attn_bench.py 2>&1 | tail -8
W2V2_ATTN_NO_XCD_REMAP=1 python3 tools/attn_bench.py 2>&1 | tail -8
