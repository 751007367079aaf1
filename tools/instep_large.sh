#!/bin/bash
set -u
R=$PWD; export TMPDIR=/tmp INSTEP_MODEL=large INSTEP_BATCH=32 INSTEP_SAMPLES=80000
cd /tmp; rm -rf /tmp/prof_isl
rocprofv3 --kernel-trace -d /tmp/prof_isl -- python3 $R/tools/gemm_instep.py run $R/gpurun_out/large_gemm_seq.json --steps 6 > $R/gpurun_out/large_instep_run.log 2>&1
DBI=$(find /tmp/prof_isl -name "*.db" | head -1)
python3 $R/tools/gemm_instep.py join $R/gpurun_out/large_gemm_seq.json $DBI $R/gpurun_out/r06_large_instep_by_shape.txt > $R/gpurun_out/large_instep_join.log 2>&1
head -40 $R/gpurun_out/r06_large_instep_by_shape.txt | cut -c1-140
tail -3 $R/gpurun_out/large_instep_join.log
