#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 3000 python3 -m pytest tests -q -m gpu > $OUT/r03_t5_all.log 2>&1; tail -25 $OUT/r03_t5_all.log
python3 bench.py --model ecapa --no-cpu-baseline > $OUT/r03_ecapa_f32.json 2>$OUT/r03_ecapa_f32.err; python3 -c "import json; d=json.load(open('$OUT/r03_ecapa_f32.json')); print('ecapa f32', d['value'], d['ms_per_step'], d['config']['final_loss'], d.get('gemm_mfma'))"
W2V2_F32_VALU=1 python3 bench.py --model ecapa --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ecapa f32 VALU gemm', d['value'], d['ms_per_step'])"
python3 bench.py --model ecapa --dtype bf16 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ecapa bf16', d['value'], d['ms_per_step'], d['config']['final_loss'])"
bash tools/ab_round.sh 2 > $OUT/r03_ab_v4.txt 2>&1; cat $OUT/r03_ab_v4.txt
