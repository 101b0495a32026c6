cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r4h.log
for k in 4 1; do
  QATZIP_AMD_SO=build/var/lib_xnomem.so QATZIP_AMD_INFLATE_K=$k timeout 60 python tools/inflate_var_run.py 4096:64 1024:64 2>&1 | tail -n 3 | sed "s/^/K=$k /" >> gpurun_out/r4h.log
done
cat gpurun_out/r4h.log
