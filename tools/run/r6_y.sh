cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so
QATZIP_AMD_INFLATE_K=8 timeout 300 python tools/prof_spec.py 4096 64 2>&1 | head -16 > gpurun_out/r6y_spec.txt
QATZIP_AMD_INFLATE_K=16 timeout 300 python tools/prof_spec.py 1024 64 2>&1 | head -16 >> gpurun_out/r6y_spec.txt
unset QATZIP_AMD_SO
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd timeout 600 bash tools/small_calls.sh > gpurun_out/small_calls.txt 2>&1
cat gpurun_out/r6y_spec.txt gpurun_out/small_calls.txt
