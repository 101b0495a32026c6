# round 6: phase A's waves on the wall clock, natural order and longest-first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so
for o in 0 1; do
  echo "== QATZIP_AMD_INFLATE_ORDER=$o"
  QATZIP_AMD_INFLATE_ORDER=$o timeout 600 python tools/prof_phaseA_timeline.py 4096
done > gpurun_out/r6q_timeline.log 2>&1
cat gpurun_out/r6q_timeline.log
