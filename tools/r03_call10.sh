#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 2400 python3 -m pytest tests/test_parity_gpu.py tests/test_surface_gpu.py tests/test_ddp_gpu.py -q -m gpu > $OUT/r03_t10.log 2>&1; tail -6 $OUT/r03_t10.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
bash tools/ab_round.sh 3 > $OUT/r03_ab_v6.txt 2>&1; cat $OUT/r03_ab_v6.txt
