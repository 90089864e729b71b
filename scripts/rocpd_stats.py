"""Per-kernel summary (calls, total / average / min / max duration, share) of a rocprofv3 run.

rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this prints the same table
`--stats` would give as CSV so that it can be committed under profiles/.
usage: python scripts/rocpd_stats.py <results.db> [> profiles/<name>_kernel_stats.csv]
"""
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         "max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks))
    rows = list(cur.execute(q))
    tot = float(sum(r[2] for r in rows)) or 1.0
    return [(r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot) for r in rows]


def main():
    print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
    for r in stats(sys.argv[1]):
        print('"%s",%d,%d,%.1f,%d,%d,%.2f' % r)


if __name__ == "__main__":
    main()
