# round 5, thirteenth GPU call: the LZ4 compressor with the test behind a match merged into the next streak's first window and the
# collision-free insert path (l4m: 7 waves per SIMD, l4m8: 8) against the round's build so far (default)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r5m_lz4.log
for v in default l4m l4m8 default l4m l4m8; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/legs_run.py lz4 1024 >> gpurun_out/r5m_lz4.log 2>&1
done
unset QATZIP_AMD_SO
cut -c1-250 gpurun_out/r5m_lz4.log
QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_l4m.so timeout 600 python -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -2
QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_l4m.so timeout 300 python tools/lz4_bench.py 1024 2>&1 | tail -1
