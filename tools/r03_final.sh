#!/bin/bash
# end-of-round evidence: full GPU suite + smoke, then the profile passes of both bench lines
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -q -m gpu > $OUT/r03_final_tests.log 2>&1; tail -5 $OUT/r03_final_tests.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh r03 > $OUT/r03_profile_round.log 2>&1; tail -40 $OUT/r03_profile_round.log
bash tools/profile_round.sh r03 ecapa > $OUT/r03_profile_round_ecapa.log 2>&1; tail -12 $OUT/r03_profile_round_ecapa.log
