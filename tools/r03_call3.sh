#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv0 or zero or adam or selective" > $OUT/r03_t3_kernels.log 2>&1; tail -5 $OUT/r03_t3_kernels.log
python3 -m pytest tests/test_parity_gpu.py tests/test_ddp_gpu.py -x -q -m gpu > $OUT/r03_t3_parity.log 2>&1; tail -5 $OUT/r03_t3_parity.log
python3 tools/gemm_shapes.py > $OUT/r03_shapes_v2.txt 2>&1; cat $OUT/r03_shapes_v2.txt
bash tools/ab_round.sh 3 > $OUT/r03_ab_v2.txt 2>&1; cat $OUT/r03_ab_v2.txt
