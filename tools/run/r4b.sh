cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r4b_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4b_pytest.log
: > gpurun_out/r4b_var.log
for k in 4 1 2 8; do
  echo "== QATZIP_AMD_INFLATE_K=$k" >> gpurun_out/r4b_var.log
  QATZIP_AMD_INFLATE_K=$k QATZIP_AMD_TRACE=1 timeout 300 python tools/inflate_var_run.py 2>&1 | grep -v "qzd_inflate_stream" | sort | uniq -c | sort -rn | head -12 >> gpurun_out/r4b_var.log
done
tail -n 6 gpurun_out/r4b_pytest.log; cat gpurun_out/r4b_var.log
