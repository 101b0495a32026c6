cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r4d_var.log
for v in xnolit xnoseq xnone xalone lr2 lr3; do
  for k in 4 1; do
  QATZIP_AMD_SO=build/var/lib_$v.so QATZIP_AMD_INFLATE_K=$k timeout 300 python tools/inflate_var_run.py 4096:64 1024:64 2>&1 | sed "s/^/K=$k /" >> gpurun_out/r4d_var.log
  done
done
cat gpurun_out/r4d_var.log
