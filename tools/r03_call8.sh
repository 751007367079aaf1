#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_ecapa_gpu.py tests/test_parity_gpu.py -q -m gpu -k "ecapa or overflow or attentive" > $OUT/r03_t8.log 2>&1; tail -6 $OUT/r03_t8.log
python3 bench.py --model ecapa --no-cpu-baseline > $OUT/r03_ecapa_f32.json 2>$OUT/r03_ecapa_f32.err; python3 -c "import json; d=json.load(open('$OUT/r03_ecapa_f32.json')); print('ecapa f32', d['value'], d['ms_per_step'], d['config']['final_loss'], d.get('gemm_mfma'))"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_e
rocprofv3 --kernel-trace --stats -d /tmp/prof_e -- python3 $R/bench.py --model ecapa --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
DB=$(find /tmp/prof_e -name "*.db" | head -1)
python3 $R/tools/prof_summary.py $DB 7 > $OUT/r03_ecapa_f32_kernel_stats.txt 2>&1; head -16 $OUT/r03_ecapa_f32_kernel_stats.txt
