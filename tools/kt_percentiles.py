"""Percentiles of kernel durations from a rocprofv3 kernel-trace CSV (stats alone hide heavy tails).  usage: kt_percentiles.py TRACE.csv"""
import csv, sys
from collections import defaultdict
import numpy as np
d = defaultdict(list)
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        d[r["Kernel_Name"].split("(")[0][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
tot = sum(sum(v) for v in d.values())
print("%-42s %7s %9s %8s %8s %8s %8s %9s %6s" % ("kernel", "calls", "mean us", "p10", "p50", "p90", "p99", "max", "share"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    a = np.array(v)
    print("%-42s %7d %9.1f %8.1f %8.1f %8.1f %8.1f %9.1f %5.1f%%" % (k, a.size, a.mean(), *np.percentile(a, [10, 50, 90, 99]), a.max(), 100 * a.sum() / tot))
