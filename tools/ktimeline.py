#!/usr/bin/env python3
"""timeline of the long kernels in a rocprofv3 kernel_trace.csv, relative to the first qzk_lz77_pull_kernel"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    n = r["Kernel_Name"][:44]; s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    if "pull" in n and t0 is None:
        t0 = s
    if t0 is None:
        continue
    if (e - s) > 100000 or "scan" in n:
        print("%-44s start %9.2f ms  dur %7.2f ms" % (n, (s - t0) / 1e6, (e - s) / 1e6))
