# round 6: K1 without the far loads behind a ring candidate that ends the chain (default) against every far candidate fetched (farall):
# launch time at 4 GiB, alternating; then FETCH_SIZE / WRITE_SIZE of one 1 GiB launch each (separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; mkdir -p $R/gpurun_out
: > $R/gpurun_out/r6s_k1.log
for v in default farall default farall default farall; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$R/build/var/lib_$v.so; fi
  timeout 300 python $R/tools/k1_var_run.py 4096 >> $R/gpurun_out/r6s_k1.log 2>&1
done
cat $R/gpurun_out/r6s_k1.log
for v in default farall; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$R/build/var/lib_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/r6s_${v}_$c
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r6s_${v}_$c -- python $R/tools/k1_var_run.py 1024 > $R/gpurun_out/r6s_${v}_$c.log 2>&1
    tail -1 $R/gpurun_out/r6s_${v}_$c.log
  done
done
cd $R; python - <<'PY' | tee gpurun_out/r6s_summary.txt
import glob, csv, collections
for v in ("default", "farall"):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("gpurun_out/r6s_%s_%s/*/*counter_collection.csv" % (v, c))[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "qzk_lz77_pull_kernel" in r["Kernel_Name"]]
        out[c] = sum(vals) / len(vals)
    print("%-8s qzk_lz77_pull_kernel per 1 GiB launch: FETCH_SIZE %.1f GiB (%.1f B per input byte)  WRITE_SIZE %.1f GiB (%.1f B per input byte)"
          % (v, out["FETCH_SIZE"] / 2**20, out["FETCH_SIZE"] * 1024 / 2**30, out["WRITE_SIZE"] / 2**20, out["WRITE_SIZE"] * 1024 / 2**30))
PY
