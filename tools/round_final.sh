#!/bin/bash
# End-of-round evidence on ONE GPU box (run through gpurun from the repo root): the full GPU test suite, smoke(), then
# the profile passes of both bench lines (tools/profile_round.sh); afterwards, in the container:
#   bash tools/collect_round.sh <tag>      # copies the results from gpurun_out/ into profiles/
# Usage: gpurun --timeout 4200 -- bash tools/round_final.sh [tag]
TAG=${1:-r06}
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -q -m gpu > $OUT/${TAG}_final_tests.log 2>&1; tail -5 $OUT/${TAG}_final_tests.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python3 tools/parity_report.py $OUT/${TAG}_parity.json 2>&1 | grep -v amdgpu | tail -14
bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1; tail -40 $OUT/${TAG}_profile_round.log
bash tools/profile_round.sh $TAG ecapa > $OUT/${TAG}_profile_round_ecapa.log 2>&1; tail -12 $OUT/${TAG}_profile_round_ecapa.log
