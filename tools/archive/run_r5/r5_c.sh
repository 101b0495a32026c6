# round 5, third GPU call: the rebuilt LZ4 block compressor (content-carrying table entries, register window, LDS-staged
# output) - parity, then config 4's rates by register budget against the round-4 kernels; the eight-rank rehearsal on one GPU.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lz4.py tests/test_gpu_golden.py -x -q > gpurun_out/r5c_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r5c_pytest.log
tail -n 4 gpurun_out/r5c_pytest.log
: > gpurun_out/r5c_lz4.log
for v in default l4o7 l4o8 r4 default; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/lz4_bench.py 1024 >> gpurun_out/r5c_lz4.log 2>&1
done
unset QATZIP_AMD_SO
timeout 300 python tools/legs_run.py lz4 1024 >> gpurun_out/r5c_lz4.log 2>&1
cat gpurun_out/r5c_lz4.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_shard.py -x -q > gpurun_out/r5c_shard.log 2>&1; echo "rc $?" >> gpurun_out/r5c_shard.log
tail -n 4 gpurun_out/r5c_shard.log
export QATZIP_AMD_RCCL_TIMEOUT=5
timeout 900 python bench.py --gpus 8 --mb 256 --members 16 --steps 1 --no-cpu > gpurun_out/r5c_bench_8ranks.json 2> gpurun_out/r5c_bench_8ranks.err; echo "bench8 rc $?"
tail -c 2500 gpurun_out/r5c_bench_8ranks.json; tail -n 5 gpurun_out/r5c_bench_8ranks.err
