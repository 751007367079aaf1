#!/bin/bash
# run the GPU suite N times in one gpurun call and report any test that does not pass every time
N=${1:-3}; OUT=$PWD/gpurun_out; mkdir -p $OUT
for i in $(seq $N); do
  timeout 2400 python3 -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/flake_$i.log 2>&1
  grep -E "passed|failed" $OUT/flake_$i.log | tail -1
  grep -E "^FAILED" $OUT/flake_$i.log
done
