#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python3 bench.py --no-cpu-baseline --no-also --pooling attentive 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('attentive', d['ms_per_step'], d['value'])"
python3 bench.py --no-cpu-baseline --no-also --model large --seconds 5 --batch 32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('large 5s b32', d['ms_per_step'], d['value'], d['model_tflops_per_gpu'])"
python3 bench.py --no-cpu-baseline --no-also --pooling first+cls 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('first+cls', d['ms_per_step'], d['value'])"
python3 bench.py --no-cpu-baseline --no-also --unfreeze-cnn 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfrozen cnn', d['ms_per_step'], d['value'])"
python3 bench.py --no-cpu-baseline --no-also --dtype f32 --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 mode', d['ms_per_step'], d['value'])"
W2V2_F32_VALU=1 python3 bench.py --no-cpu-baseline --no-also --dtype f32 --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 mode VALU gemm', d['ms_per_step'], d['value'])"
