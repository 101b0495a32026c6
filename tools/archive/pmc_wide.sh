#!/bin/bash
# Fabric traffic of the workgroup-per-chunk parse (QATZIP_AMD_K1=wide) beside K1's on the same 512 MiB: two rocprofv3 --pmc
# passes each (FETCH_SIZE and WRITE_SIZE separately, as the guide prescribes), per-kernel sums printed.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in wide pull; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcw_${mode}_$ctr
    QATZIP_AMD_K1=$mode timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_${mode}_$ctr -- python $R/tools/k1_var_run.py 512 > $R/gpurun_out/pmcw_${mode}_$ctr.log 2>&1
  done
done
python - <<PY
import csv, glob, collections
R = "$R"
for mode in ("wide", "pull"):
    print("== QATZIP_AMD_K1=%s, 512 MiB of the bench data, 4 calls (tools/k1_var_run.py 512): KiB per launch, averaged" % mode)
    tot = collections.defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob("%s/gpurun_out/pmcw_%s_%s/*/*counter_collection.csv" % (R, mode, ctr))
        if not fs:
            print("   (no %s collection)" % ctr); continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == ctr:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            tot[k][ctr] = sum(v) / len(v); tot[k]["n"] = len(v)
    for k, v in sorted(tot.items()):
        if k.startswith("qzk_"):
            print("   %-42s launches %3d  FETCH_SIZE %12.0f KiB  WRITE_SIZE %12.0f KiB" % (k[:42], v.get("n", 0), v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
PY
