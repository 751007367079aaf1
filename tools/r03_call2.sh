#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
FAMILIES=0,1,2,3,4 TRIALS=3 python3 tools/gemm_shapes.py > $OUT/r03_shapes_families.txt 2>&1
./tools/probes/store_pattern_probe > $OUT/r03_store_probe.txt 2>&1
for u in 1 2 4; do for nb in 8192 2048 1024; do W2V2_ADAM_U=$u W2V2_ADAM_BLOCKS=$nb python3 tools/adam_bench.py 2>/dev/null | tail -1; done; done > $OUT/r03_adam.txt
cat $OUT/r03_shapes_families.txt $OUT/r03_store_probe.txt $OUT/r03_adam.txt
