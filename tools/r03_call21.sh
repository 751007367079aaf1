python3 tools/dbg_zero.py 2>&1 | tail -40
