#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 3000 python3 -m pytest tests -q -m gpu > $OUT/r03_t7_all.log 2>&1; tail -12 $OUT/r03_t7_all.log
python3 bench.py --model ecapa --no-cpu-baseline > $OUT/r03_ecapa_f32.json 2>$OUT/r03_ecapa_f32.err; python3 -c "import json; d=json.load(open('$OUT/r03_ecapa_f32.json')); print('ecapa f32', d['value'], d['ms_per_step'], d['config']['final_loss'], d.get('gemm_mfma'))"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_is
rocprofv3 --kernel-trace -d /tmp/prof_is -- python3 $R/tools/gemm_instep.py run $OUT/r03_seq_v2.json --steps 6 > $OUT/r03_instep_run2.log 2>&1
DB=$(find /tmp/prof_is -name "*.db" | head -1)
python3 $R/tools/gemm_instep.py join $OUT/r03_seq_v2.json $DB $OUT/r03_instep_v2.txt > $OUT/r03_instep_join2.log 2>&1
cd $R; head -20 $OUT/r03_instep_v2.txt; tail -3 $OUT/r03_instep_join2.log
bash tools/ab_round.sh 2 > $OUT/r03_ab_v5.txt 2>&1; cat $OUT/r03_ab_v5.txt
