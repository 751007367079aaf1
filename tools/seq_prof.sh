#!/bin/bash
# One traced step in stream order (tools/step_sequence.py) -> gpurun_out/<name>.txt.   Usage: bash tools/seq_prof.sh <name> [bench args]
N=${1:-seq}; shift
R=$PWD; export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/seq && rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -- python3 $R/bench.py --no-cpu-baseline --no-also --no-eer --steps 6 --warmup 3 "$@" > /tmp/seq.log 2>&1
python3 $R/tools/step_sequence.py /tmp/seq 2 > $R/gpurun_out/$N.txt; tail -1 $R/gpurun_out/$N.txt
