#!/bin/bash
# kernel-time table of BASELINE configs[2]'s per-GPU share (w2v2-base + attentive statistics pooling, 66 x 3 s)
set -u
R=$PWD; export TMPDIR=/tmp W2V2_BENCH_NO_FAMILY_PASS=1
cd /tmp; rm -rf /tmp/prof_at
rocprofv3 --kernel-trace --stats -d /tmp/prof_at -- python3 $R/bench.py --pooling attentive --no-cpu-baseline --no-also --no-eer --steps 6 --warmup 3 > $R/gpurun_out/att_prof.log 2>&1
DB=$(find /tmp/prof_at -name "*.db" | head -1)
python3 $R/tools/prof_summary.py $DB 6 --steady adam_kernel > $R/gpurun_out/r06_attentive_b66_kernel_stats.txt 2>&1
grep -n "asp\|skinny\|bn_\|total kernel" $R/gpurun_out/r06_attentive_b66_kernel_stats.txt | head -30
