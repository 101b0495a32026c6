# round 6: phase A with eight waves a CU (its group bookkeeping moved into the root rows' spare bytes: LDS 20928 -> 20480 B a wave)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6r_inflate.log
for i in 1 2 3; do timeout 600 python tools/inflate_var_run.py 4096:64 1024:64 256:64 1024:128 1024:16 >> gpurun_out/r6r_inflate.log 2>&1; done
QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so timeout 600 python tools/prof_phaseA_timeline.py 4096 >> gpurun_out/r6r_inflate.log 2>&1
cat gpurun_out/r6r_inflate.log
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
