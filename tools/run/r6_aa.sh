# round 6: what K2 in the wave costs the 4 x 5 kernel (-DQZK_K1_NOK2: parse + CRC only, no stream)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6aa_k1.log
for v in default nok2 default nok2; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6aa_k1.log 2>&1
done
cat gpurun_out/r6aa_k1.log
