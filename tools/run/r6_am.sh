# round 6: compiler knobs on the whole library, K1's launch time (max-ilp / max-memory-clause scheduling, -O2, -Os)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6am_k1.log
for v in default ilp memcl o2 os default ilp memcl o2 os; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6am_k1.log 2>&1
done
cat gpurun_out/r6am_k1.log
