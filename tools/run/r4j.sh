cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r4j.log
for k in 4; do
  QATZIP_AMD_TRACE=1 QATZIP_AMD_INFLATE_K=$k timeout 90 python tools/inflate_var_run.py 4096:64 1024:64 1024:128 1024:16 64:64 2>&1 | grep -v "qzd_inflate_stream" | sort | uniq -c | sort -rn | sed "s/default  */K=$k  /" >> gpurun_out/r4j.log
done
cat gpurun_out/r4j.log
timeout 300 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
