# round 5, fifteenth GPU call: phase A with the reach limit, the taken-back pieces and the whole-block guess (default; reach4 /
# reach16: QZK_SPEC_REACH) against the kernel before them (pa0), the per-segment times, the decode tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r5o_phaseA.log
for v in pa0 default reach4 reach16 pa0 default; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/inflate_var_run.py 64:64 256:64 1024:64 4096:64 1024:128 1024:16 >> gpurun_out/r5o_phaseA.log 2>&1
done
QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so timeout 300 python tools/prof_phaseA_tail.py 64 256 1024 4096 > gpurun_out/phaseA_tail_o.txt 2>&1
unset QATZIP_AMD_SO
timeout 900 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu > gpurun_out/r5o_tests.log 2>&1
tail -n 3 gpurun_out/r5o_tests.log
cat gpurun_out/r5o_phaseA.log | cut -c1-170
tail -n 14 gpurun_out/phaseA_tail_o.txt
