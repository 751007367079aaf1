python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'], d['loss_scale'])"; done
