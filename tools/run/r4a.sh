cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q > gpurun_out/r4a_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4a_pytest.log
: > gpurun_out/r4a_var.log
for v in r3 n4 n6 n8 d6 t1k; do
  QATZIP_AMD_SO=build/var/lib_$v.so timeout 300 python tools/inflate_var_run.py >> gpurun_out/r4a_var.log 2>&1
done
tail -n 5 gpurun_out/r4a_pytest.log; cat gpurun_out/r4a_var.log
