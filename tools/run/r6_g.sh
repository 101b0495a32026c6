# round 6: what the LDS entry cache (and asking ahead) do to K1's fabric traffic: FETCH_SIZE / WRITE_SIZE (separate passes) of one 1 GiB
# launch of qzk_lz77_pull_kernel for the default build, the cache alone and cache + asked ahead
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; mkdir -p $R/gpurun_out
for v in default cache pf; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$R/build/var/lib_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r6g_${v}_$c -- python $R/tools/k1_var_run.py 1024 > $R/gpurun_out/r6g_${v}_$c.log 2>&1
    tail -1 $R/gpurun_out/r6g_${v}_$c.log
  done
done
cd $R; python - <<'PY'
import glob, csv, collections
for v in ("default", "cache", "pf"):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("gpurun_out/r6g_%s_%s/*/*counter_collection.csv" % (v, c))[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "qzk_lz77_pull_kernel" in r["Kernel_Name"]]
        out[c] = sum(vals) / len(vals)
    print("%-8s qzk_lz77_pull_kernel per 1 GiB launch: FETCH_SIZE %.1f GiB (%.1f B per input byte)  WRITE_SIZE %.1f GiB (%.1f B per input byte)"
          % (v, out["FETCH_SIZE"] / 2**20, out["FETCH_SIZE"] * 1024 / 2**30, out["WRITE_SIZE"] / 2**20, out["WRITE_SIZE"] * 1024 / 2**30))
PY
