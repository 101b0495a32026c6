cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6ae_k1.log
for v in noslot3 slot3_64 slot3_32 slot3_128b noslot3 slot3_64 slot3_32 slot3_128b; do
  export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6ae_k1.log 2>&1
done
cat gpurun_out/r6ae_k1.log
