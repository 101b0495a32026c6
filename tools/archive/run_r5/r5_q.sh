# round 5, seventeenth GPU call: reach 4 for a lane nobody vouches for and 8 for one on the true stream (default) against 4 for
# all (reach4) and 4 / 16, 4 / 6 (c16, c6); 2 GiB at K = 8
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r5q_phaseA.log
for v in default reach4 c16 c6 default reach4; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/inflate_var_run.py 64:64 256:64 1024:64 2048:64 4096:64 1024:128 1024:16 512:512 >> gpurun_out/r5q_phaseA.log 2>&1
done
unset QATZIP_AMD_SO
echo "K=8" >> gpurun_out/r5q_phaseA.log
QATZIP_AMD_INFLATE_K=8 timeout 300 python tools/inflate_var_run.py 2048:64 4096:64 1024:16 >> gpurun_out/r5q_phaseA.log 2>&1
cut -c1-170 gpurun_out/r5q_phaseA.log
