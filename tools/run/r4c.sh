cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r4c_var.log
for k in 4 8 1; do
  echo "== QATZIP_AMD_INFLATE_K=$k" >> gpurun_out/r4c_var.log
  QATZIP_AMD_INFLATE_K=$k QATZIP_AMD_TRACE=1 timeout 300 python tools/inflate_var_run.py 2>&1 | grep -v "qzd_inflate_stream" | sort | uniq -c | sort -rn | head -12 >> gpurun_out/r4c_var.log
done
cat gpurun_out/r4c_var.log
