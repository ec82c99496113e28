"""kernel durations out of a rocprofv3 results .db (rocprofv3 --kernel-trace -d DIR -o NAME)"""
import collections, glob, sqlite3, sys
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    ks = dict(c.execute("select id, kernel_name from %s" % sym).fetchall())
    agg = collections.defaultdict(list)
    for kid, st, en in c.execute("select kernel_id, start, end from %s" % disp):
        agg[ks.get(kid, kid)].append(en - st)
    print("==", db)
    for k, v in sorted(agg.items(), key=lambda t: -sum(t[1])):
        if "sd" in str(k) or "rocclr" in str(k):
            print("%9.1f us avg  x%5d  %s" % (sum(v) / len(v) / 1e3, len(v), str(k)[:110]))
