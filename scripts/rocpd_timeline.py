"""Timeline of the LAST theta-step evaluation in a rocprofv3 run of scripts/time_mvn.py: every kernel dispatch between
the last two k_build_dense launches... (start relative to the evaluation, duration, gap to the previous end on any queue).
usage: python scripts/rocpd_timeline.py <results.db> [max_rows]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
builds = [i for i, r in enumerate(rows) if "k_build_dense" in r[0]]
lo = builds[-1]
sel = rows[lo:]
t0 = sel[0][1]
prev_end = t0
short = lambda n: ("leaf" if "potrf_leaf" in n else "dl<%s>" % n.split("dgemm_dl_kernelILi")[1][:13] if "dgemm_dl_kernel" in n else n.split("mcml")[1][:24] if "mcml" in n else n[:24])
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for n, s, e, q in sel[:lim]:
    print("%9.1f us  +%7.1f  dur %7.1f  q%-3s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, short(n)))
    prev_end = max(prev_end, e)
print("total %.1f us" % ((max(r[2] for r in sel) - t0) / 1e3))
