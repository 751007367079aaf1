#!/bin/bash
# copy the evidence tools/profile_round.sh left in gpurun_out/ into profiles/ under the names the docs and tests cite
TAG=${1:-r06}
G=gpurun_out; P=profiles
cp $G/${TAG}_bench_line.json $P/${TAG}_bench_line.json
cp $G/${TAG}_kernel_stats.txt $P/${TAG}_bench_b66_kernel_stats.txt
cp $G/${TAG}_pmc_counters.json $P/${TAG}_pmc_counters.json
cp $G/${TAG}_instep_by_shape.txt $P/${TAG}_instep_by_shape_after.txt
cp $G/${TAG}_ecapa_bench_line.json $P/${TAG}_ecapa_bench_line.json
cp $G/${TAG}_ecapa_kernel_stats.txt $P/${TAG}_ecapa_b66_kernel_stats.txt
cp $G/${TAG}_ecapa_pmc_counters.json $P/${TAG}_ecapa_pmc_counters.json
cp $G/${TAG}_parity.json $P/${TAG}_parity.json
