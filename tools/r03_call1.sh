#!/bin/bash
# round-3 GPU call 1: in-step per-shape table + stand-alone shapes with epilogues under the kernel A/B switches
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_is
rocprofv3 --kernel-trace -d /tmp/prof_is -- python3 $R/tools/gemm_instep.py run $OUT/r03_seq_base.json --steps 6 > $OUT/r03_instep_run.log 2>&1
DB=$(find /tmp/prof_is -name "*.db" | head -1)
python3 $R/tools/gemm_instep.py join $OUT/r03_seq_base.json $DB $OUT/r03_instep_base.txt > /dev/null 2>$OUT/r03_instep_join.err
python3 $R/tools/prof_summary.py $DB 9 > $OUT/r03_instep_kernel_stats.txt 2>&1
cd $R
python3 tools/gemm_shapes.py --hipblaslt > $OUT/r03_shapes_default.txt 2>&1
W2V2_NO_GLDS4=1 python3 tools/gemm_shapes.py > $OUT/r03_shapes_glds3only.txt 2>&1
W2V2_NO_GEMM_PH=1 python3 tools/gemm_shapes.py f > $OUT/r03_shapes_glds4.txt 2>&1
python3 bench.py --no-cpu-baseline > $OUT/r03_bench_base.json 2>$OUT/r03_bench_base.err
tail -c 400 $OUT/r03_bench_base.json
cat $OUT/r03_instep_base.txt
