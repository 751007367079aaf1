import sqlite3, sys, glob
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t == "counters_collection"]
    q = ("select k.name, c.counter_name, avg(c.value), count(*) from counters_collection c "
         "join kernels k on k.dispatch_id = c.dispatch_id group by k.name, c.counter_name")
    try:
        rows = list(cur.execute(q))
    except Exception as e:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        print("cols", cols); raise
    for n, cn, val, cnt in rows:
        if "wgrad" in n or "gemm16_ring" in n:
            print(f"{n[:50]:50s} {cn:28s} {val:16.1f} n={cnt}")
