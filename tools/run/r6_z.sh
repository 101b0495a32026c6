cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6z_inflate.log
for v in default hdrcall default hdrcall; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 600 python tools/inflate_var_run.py 4096:64 2048:64 1024:64 256:64 64:64 1024:16 >> gpurun_out/r6z_inflate.log 2>&1
done
cat gpurun_out/r6z_inflate.log
