# round 5, second GPU call: the LZ4 frame decoder on the batch engine and the block compressor without its store waits -
# parity (LZ4 tests, goldens, the API), then config 4's rates against the round-4 kernels in one session.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lz4.py tests/test_gpu_golden.py -x -q > gpurun_out/r5b_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r5b_pytest.log
tail -n 4 gpurun_out/r5b_pytest.log
timeout 900 python -m pytest tests/test_gpu_api.py -x -q -k "lz4 or LZ4" >> gpurun_out/r5b_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r5b_pytest.log
tail -n 3 gpurun_out/r5b_pytest.log
: > gpurun_out/r5b_lz4.log
for v in default r4 default r4; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/lz4_bench.py 1024 >> gpurun_out/r5b_lz4.log 2>&1
done
unset QATZIP_AMD_SO
timeout 300 python tools/legs_run.py lz4 1024 >> gpurun_out/r5b_lz4.log 2>&1
cat gpurun_out/r5b_lz4.log
