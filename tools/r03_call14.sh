#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python3 tools/wgrad_fill.py 2>&1 | tee $OUT/r03_wgrad_fill.txt
python3 tools/aten_in_step.py > $OUT/r03_aten_in_step.txt 2>$OUT/r03_aten_in_step.err; tail -5 $OUT/r03_aten_in_step.err; cat $OUT/r03_aten_in_step.txt
