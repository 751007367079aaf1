#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 3000 python3 -m pytest tests -q -m gpu > $OUT/r03_t6_all.log 2>&1; tail -12 $OUT/r03_t6_all.log
python3 bench.py --model ecapa --no-cpu-baseline > $OUT/r03_ecapa_f32.json 2>$OUT/r03_ecapa_f32.err; python3 -c "import json; d=json.load(open('$OUT/r03_ecapa_f32.json')); print('ecapa f32', d['value'], d['ms_per_step'], d['config']['final_loss'], d.get('gemm_mfma'))"
W2V2_F32_VALU=1 python3 bench.py --model ecapa --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ecapa f32 VALU gemm', d['value'], d['ms_per_step'])"
python3 bench.py --model ecapa --dtype bf16 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ecapa bf16', d['value'], d['ms_per_step'], d['config']['final_loss'])"
for i in 1 2; do for v in 1 0; do if [ $v = 1 ]; then export W2V2_NO_WGRAD_ORDER=1; else unset W2V2_NO_WGRAD_ORDER; fi; python3 bench.py --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad order off=$v', d['ms_per_step'], d['value'])"; done; done
unset W2V2_NO_WGRAD_ORDER
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_is
rocprofv3 --kernel-trace -d /tmp/prof_is -- python3 $R/tools/gemm_instep.py run $OUT/r03_seq_v2.json --steps 6 > $OUT/r03_instep_run2.log 2>&1
DB=$(find /tmp/prof_is -name "*.db" | head -1)
python3 $R/tools/gemm_instep.py join $OUT/r03_seq_v2.json $DB $OUT/r03_instep_v2.txt > /dev/null 2>&1
python3 $R/tools/prof_summary.py $DB 9 > $OUT/r03_kernel_stats_v2.txt 2>&1
cd $R; head -32 $OUT/r03_kernel_stats_v2.txt; head -16 $OUT/r03_instep_v2.txt
