# round 6: K1 with a third slot table (128 entries under a mix of the hash's bits) against two
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6ad_k1.log
for v in default noslot3 default noslot3 default noslot3; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6ad_k1.log 2>&1
done
cat gpurun_out/r6ad_k1.log
