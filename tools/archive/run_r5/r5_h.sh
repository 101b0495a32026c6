# round 5, eighth GPU call: large segments - the first block's guess (12 / 20 / 32 KB of output)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r5h_inflate.log
for v in default fk12 fk32; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/inflate_var_run.py 1024:512 1024:256 128:512 64:256 1024:128 >> gpurun_out/r5h_inflate.log 2>&1
done
unset QATZIP_AMD_SO
cat gpurun_out/r5h_inflate.log
