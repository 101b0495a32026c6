#!/usr/bin/env python3
"""Developer tool: print the per-kernel averages of the rocprofv3 --pmc passes tools/pmc_any.sh left under gpurun_out/<tag>[A-E]."""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1]
filt = sys.argv[2:] or ["qzk_"]
for d in "ABCDE":
    for f in sorted(glob.glob("gpurun_out/%s%s/*/*counter_collection.csv" % (tag, d)), key=os.path.getmtime)[-1:]:     # the newest pass only
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(x in k for x in filt):
                agg[(k[:34], r["Counter_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in sorted(agg.items()):
            print("%s %-34s %-30s n=%-3d avg=%-11.4g dur_ms=%.2f" % (d, k[0], k[1], len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v) / 1e6))
