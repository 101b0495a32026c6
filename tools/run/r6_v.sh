cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6v_inflate.log
for v in "socc3 8" "socc4 8" "socc4 16" "socc3 16" "socc3 8" "socc4 8" "socc4 16"; do
  set -- $v
  if [ $1 = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$1.so; fi
  echo "== $1 K=$2" >> gpurun_out/r6v_inflate.log
  QATZIP_AMD_INFLATE_K=$2 timeout 600 python tools/inflate_var_run.py 4096:64 2048:64 >> gpurun_out/r6v_inflate.log 2>&1
done
unset QATZIP_AMD_SO
echo "== default (own K rule)" >> gpurun_out/r6v_inflate.log
timeout 600 python tools/inflate_var_run.py 4096:64 2048:64 >> gpurun_out/r6v_inflate.log 2>&1
cat gpurun_out/r6v_inflate.log
