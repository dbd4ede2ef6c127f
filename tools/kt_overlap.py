"""Do the kernels of a rocprofv3 kernel trace run side by side?  For the kernels whose name starts with one of the given prefixes:
sum of durations, length of the union of their intervals, and the share of the union during which two or more were running.
usage: kt_overlap.py TRACE.csv [prefix ...]   (default prefix: k_pen_)"""
import csv, sys
pre = tuple(sys.argv[2:]) or ("k_pen_",)
iv, queues = [], set()
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        if r["Kernel_Name"].startswith(pre):
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            queues.add(r.get("Queue_Id", "?"))
ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
depth, last, busy, multi = 0, None, 0, 0
for t, d in ev:
    if depth > 0: busy += t - last
    if depth > 1: multi += t - last
    depth += d; last = t
tot = sum(b - a for a, b in iv)
print("kernels %d on queues %s: sum of durations %.1f ms, union %.1f ms, two or more running during %.1f %% of the union" %
      (len(iv), sorted(queues), tot * 1e-6, busy * 1e-6, 100.0 * multi / max(busy, 1)))
