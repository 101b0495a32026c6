# round 6: decode scratch by need + hand-back area: rates and scratch per output byte at the bench's shapes, other data
# kinds (what the K-lane kernel hands back), then the decode-side tests; LZ4 BD check; the widened mark tag
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
export QATZIP_AMD_TRACE=1
timeout 600 python tools/inflate_var_run.py 4096:64 1024:64 256:64 64:64 1024:128 1024:16 1024:512 > gpurun_out/r6i_inflate.log 2>&1
for k in text records runs rand lzmix; do SWEEP_KIND=$k timeout 300 python tools/inflate_var_run.py 1024:64 256:16 >> gpurun_out/r6i_inflate.log 2>&1; done
unset QATZIP_AMD_TRACE
grep -v "^\[two_phase\] K=1\b" gpurun_out/r6i_inflate.log | tail -40
timeout 1500 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_lz4.py tests/test_gpu_api.py tests/test_gpu_golden.py -x -q > gpurun_out/r6i_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r6i_pytest.log
tail -n 5 gpurun_out/r6i_pytest.log
