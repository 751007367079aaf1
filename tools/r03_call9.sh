#!/bin/bash
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
bash tools/profile_round.sh r03 ecapa > gpurun_out/r03_profile_round_ecapa.log 2>&1
cp profiles/r03_pmc_counters.json profiles/r03_ecapa_pmc_counters.json gpurun_out/ 2>/dev/null
tail -c 1500 gpurun_out/r03_bench_line.json; echo; head -24 gpurun_out/r03_kernel_stats.txt; head -22 gpurun_out/r03_instep_by_shape.txt; tail -c 800 gpurun_out/r03_ecapa_bench_line.json
