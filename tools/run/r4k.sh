cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
export QATZIP_AMD_SO=build/var/lib_sprof.so
(timeout 90 python tools/prof_spec.py 64 64 | grep -v Warning | head -60) > gpurun_out/r4k.log 2>&1
(QATZIP_AMD_INFLATE_K=16 timeout 90 python tools/prof_spec.py 64 64 | grep -v Warning | head -60) >> gpurun_out/r4k.log 2>&1
cat gpurun_out/r4k.log
