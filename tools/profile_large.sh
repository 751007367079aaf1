#!/bin/bash
# kernel-time table of BASELINE configs[3]'s per-GPU share (wav2vec2-large, 5 s clips, 32 utterances), steady-state window
set -u
R=$PWD; export TMPDIR=/tmp W2V2_BENCH_NO_FAMILY_PASS=1
cd /tmp; rm -rf /tmp/prof_lg
rocprofv3 --kernel-trace --stats -d /tmp/prof_lg -- python3 $R/bench.py --model large --seconds 5 --batch 32 --no-cpu-baseline --no-also --no-eer --steps 6 --warmup 3 > $R/gpurun_out/large_prof.log 2>&1
DB=$(find /tmp/prof_lg -name "*.db" | head -1)
python3 $R/tools/prof_summary.py $DB 6 --steady adam_kernel > $R/gpurun_out/r06_large_5s_b32_kernel_stats.txt 2>&1
head -40 $R/gpurun_out/r06_large_5s_b32_kernel_stats.txt
tail -c 300 $R/gpurun_out/large_prof.log
