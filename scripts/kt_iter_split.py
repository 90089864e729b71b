"""split a rocprofv3 kernel trace (csv) of an MCML run into iterations (one k_cm_init / k_hmc_init per hmc_sample) and
print, per iteration: span, GPU-busy time, idle time, launches, and the largest gaps -- to tell slow kernels from idle GPU"""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_cm_init" in r[2] or "k_hmc_init" in r[2]]
starts.append(len(rows))
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    span = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1))
    big = [g for g in gaps if g[0] > 100]
    print("iter@%d: n=%d span=%.1f ms busy=%.1f ms idle=%.1f ms; gaps>100us: %d (%.1f ms); median gap %.1f us" %
          (a, len(seg), span, busy, span - busy, len(big), sum(g[0] for g in big) / 1e3, gaps[len(gaps) // 2][0]))
    for g in gaps[-3:]:
        print("     gap %.0f us after %s before %s" % g)
