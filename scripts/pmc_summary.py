"""Average of every PMC counter per kernel from a rocprofv3 --pmc ... --output-format csv run.
usage: python scripts/pmc_summary.py <counter_collection.csv> [kernel-substring ...]"""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
pats = sys.argv[2:]
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if pats and not any(p in k for p in pats):
        continue
    a = acc[k][row["Counter_Name"]]
    a[0] += float(row["Counter_Value"]); a[1] += 1
print('"Kernel","Counter","Launches","AveragePerLaunch"')
for k in sorted(acc):
    for cn in sorted(acc[k]):
        s, n = acc[k][cn]
        print('"%s","%s",%d,%.1f' % (k[:140], cn, n, s / max(n, 1)))
