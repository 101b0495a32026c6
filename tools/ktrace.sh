#!/bin/bash
# usage: ktrace.sh <tag> <python tool + args...>  -- rocprofv3 --kernel-trace --stats of the command, summary to gpurun_out/<tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python $R/"$@" > $R/gpurun_out/$tag.log 2>&1
grep -v "^[WE]2026" $R/gpurun_out/$tag.log | tail -4
python - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/$tag/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %5s avg %10.1f us  total %8.2f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
