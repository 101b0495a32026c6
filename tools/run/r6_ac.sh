# round 6: K1's own counters on the round's shape (4 waves x 5 workgroups a CU), one 1 GiB launch per pass: which unit of the CU is the busy one
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
bash tools/pmc_any.sh k1c tools/k1_var_run.py 1024 > gpurun_out/r6ac_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/k1cF -- python $R/tools/k1_var_run.py 1024 > $R/gpurun_out/k1cF.log 2>&1
timeout 150 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_IFETCH --kernel-trace --output-format csv -d $R/gpurun_out/k1cG -- python $R/tools/k1_var_run.py 1024 > $R/gpurun_out/k1cG.log 2>&1
cd $R
python tools/pmc_show.py k1c qzk_lz77_pull > gpurun_out/r6ac_counters.txt 2>&1
python - <<'PY' >> gpurun_out/r6ac_counters.txt
import collections, csv, glob
for d in "FG":
    for f in glob.glob("gpurun_out/k1c%s/*/*counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "qzk_lz77_pull" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in sorted(agg.items()):
            print("%s qzk_lz77_pull_kernel %-30s n=%-3d avg=%-11.4g dur_ms=%.2f" % (d, k, len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v) / 1e6))
PY
cat gpurun_out/r6ac_counters.txt; tail -3 gpurun_out/k1cF.log gpurun_out/k1cG.log | cut -c1-200
