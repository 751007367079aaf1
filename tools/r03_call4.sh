#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -q -m gpu -x --deselect tests/test_parity_gpu.py::test_b66_fp16_training_steps_of_the_benchmarked_configuration > $OUT/r03_t4_all.log 2>&1; tail -15 $OUT/r03_t4_all.log
timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -m gpu -k "b66_fp16" > $OUT/r03_t4_b66.log 2>&1; tail -8 $OUT/r03_t4_b66.log
bash tools/ab_round.sh 2 > $OUT/r03_ab_v3.txt 2>&1; cat $OUT/r03_ab_v3.txt
