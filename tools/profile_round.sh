#!/bin/bash
# Produce the committed evidence of a round on the GPU box (run through gpurun from the repo root):
#   kernel-time table, HBM-traffic and matrix-pipe counters of the default bench command.
# Usage: bash tools/profile_round.sh <tag> [ecapa]
#   -> gpurun_out/<tag>_{kernel_stats.txt,pmc_counters.json,bench_line.json}   (with `ecapa`: <tag>_ecapa_*, configs[4])
set -u
TAG=${1:-r06}
MODEL=""
if [ "${2:-}" = "ecapa" ]; then TAG=${TAG}_ecapa; MODEL="--model ecapa"; fi
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export W2V2_BENCH_NO_FAMILY_PASS=1      # profiled runs: only the timed steps (no second, event-instrumented pass)
BENCH="python3 $PWD/bench.py --no-cpu-baseline --no-also $MODEL"
cd /tmp
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w /tmp/prof_m /tmp/prof_v
rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -- $BENCH --steps 8 --warmup 3 > $OUT/${TAG}_prof_ks.log 2>&1
DB=$(find /tmp/prof_ks -name "*.db" | head -1)
# the last 8 steps only (steady state: start-up copies / fills / weight packing left out), delimited by the
# once-per-step adam_kernel
python3 $OLDPWD/tools/prof_summary.py $DB 8 --steady adam_kernel > $OUT/${TAG}_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -- $BENCH --steps 2 --warmup 1 > $OUT/${TAG}_prof_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -- $BENCH --steps 2 --warmup 1 > $OUT/${TAG}_prof_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_m -- $BENCH --steps 2 --warmup 1 > $OUT/${TAG}_prof_m.log 2>&1
rm -rf /tmp/prof_v
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d /tmp/prof_v -- $BENCH --steps 2 --warmup 1 > $OUT/${TAG}_prof_v.log 2>&1
python3 $OLDPWD/tools/pmc_counters.py /tmp/prof_f /tmp/prof_w /tmp/prof_m /tmp/prof_v > $OUT/${TAG}_pmc_counters.json 2> $OUT/${TAG}_pmc_err.log
cd $OLDPWD
# bench.py / tools/ecapa_bench.py read the counters from profiles/: refresh that copy BEFORE the line is produced, so the
# line's traffic / mfma_busy fields come from counters taken on exactly these kernel sources (pmc_stale: false)
cp $OUT/${TAG}_pmc_counters.json profiles/${TAG}_pmc_counters.json
if [ -z "$MODEL" ]; then
  # per-shape in-step table of the matrix-core launches (tools/gemm_instep.py)
  cd /tmp; rm -rf /tmp/prof_is
  rocprofv3 --kernel-trace -d /tmp/prof_is -- python3 $OLDPWD/tools/gemm_instep.py run $OUT/${TAG}_gemm_seq.json --steps 6 > $OUT/${TAG}_instep_run.log 2>&1
  DBI=$(find /tmp/prof_is -name "*.db" | head -1)
  python3 $OLDPWD/tools/gemm_instep.py join $OUT/${TAG}_gemm_seq.json $DBI $OUT/${TAG}_instep_by_shape.txt > $OUT/${TAG}_instep_join.log 2>&1
  cd $OLDPWD
fi
unset W2V2_BENCH_NO_FAMILY_PASS
python3 bench.py $MODEL > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_err.log
tail -c 600 $OUT/${TAG}_bench_line.json
head -30 $OUT/${TAG}_kernel_stats.txt
