# round 5, first GPU call: the rebuilt phase B (qzk_lz_batch.h) on the hardware - parity, then its time against the round-4
# kernel and its own variants in ONE session; K1 with K2 compiled out; a bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r5a_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r5a_pytest.log
tail -n 3 gpurun_out/r5a_pytest.log
: > gpurun_out/r5a_inflate.log
for v in default r4 o7 o8 lim2k lim4k; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/inflate_var_run.py 4096:64 1024:64 256:64 64:64 1024:128 1024:16 >> gpurun_out/r5a_inflate.log 2>&1
done
unset QATZIP_AMD_SO
cat gpurun_out/r5a_inflate.log
: > gpurun_out/r5a_k1.log
for v in default nok2 default nok2; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r5a_k1.log 2>&1
done
unset QATZIP_AMD_SO
cat gpurun_out/r5a_k1.log
timeout 600 python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; tail -c 1500 gpurun_out/r5a_bench.json
