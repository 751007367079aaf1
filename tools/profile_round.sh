#!/bin/bash
# Produce the committed evidence of a round on the GPU box (run through gpurun from the repo root):
#   kernel-time table, HBM-traffic and matrix-pipe counters of the default bench command.
# Usage: bash tools/profile_round.sh <tag> [ecapa]
#   -> gpurun_out/<tag>_{kernel_stats.txt,pmc_counters.json,bench_line.json}   (with `ecapa`: <tag>_ecapa_*, configs[4])
set -u
TAG=${1:-r03}
MODEL=""
if [ "${2:-}" = "ecapa" ]; then TAG=${TAG}_ecapa; MODEL="--model ecapa"; fi
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export W2V2_BENCH_NO_FAMILY_PASS=1      # profiled runs: only the timed steps (no second, event-instrumented pass)
BENCH="python3 $PWD/bench.py --no-cpu-baseline --no-also $MODEL"
cd /tmp
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w /tmp/prof_m
rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -- $BENCH --steps 8 --warmup 3 > $OUT/${TAG}_prof_ks.log 2>&1
DB=$(find /tmp/prof_ks -name "*.db" | head -1)
python3 $OLDPWD/tools/prof_summary.py $DB 11 > $OUT/${TAG}_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -- $BENCH --steps 2 --warmup 1 > $OUT/${TAG}_prof_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -- $BENCH --steps 2 --warmup 1 > $OUT/${TAG}_prof_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_m -- $BENCH --steps 2 --warmup 1 > $OUT/${TAG}_prof_m.log 2>&1
python3 $OLDPWD/tools/pmc_counters.py /tmp/prof_f /tmp/prof_w /tmp/prof_m > $OUT/${TAG}_pmc_counters.json 2> $OUT/${TAG}_pmc_err.log
cd $OLDPWD
unset W2V2_BENCH_NO_FAMILY_PASS
python3 bench.py $MODEL > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_err.log
tail -c 600 $OUT/${TAG}_bench_line.json
head -30 $OUT/${TAG}_kernel_stats.txt
