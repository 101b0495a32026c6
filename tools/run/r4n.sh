cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
QATZIP_AMD_INFLATE_K=4 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt4 -- python $R/tools/inflate_var_run.py 4096:64 > $R/gpurun_out/kt4.log 2>&1
cd $R; tail -n 2 gpurun_out/kt4.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt4/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
sel = [r for r in rows if "inflate" in r["Kernel_Name"] or "resolve" in r["Kernel_Name"] or "marker" in r["Kernel_Name"] or "crc" in r["Kernel_Name"]]
last = sel[-12:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-40s start %8.3f ms  dur %8.3f ms" % (r["Kernel_Name"].split("(")[0][-40:], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
