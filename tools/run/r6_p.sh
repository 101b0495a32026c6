# round 6: phase A's launch places by compressed length, longest first (QATZIP_AMD_INFLATE_ORDER=0: the segments' own order), alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6p_inflate.log
for o in 1 0 1 0 1 0; do
  echo "== QATZIP_AMD_INFLATE_ORDER=$o" >> gpurun_out/r6p_inflate.log
  QATZIP_AMD_INFLATE_ORDER=$o timeout 600 python tools/inflate_var_run.py 4096:64 1024:64 1024:128 1024:16 >> gpurun_out/r6p_inflate.log 2>&1
done
echo "== default" >> gpurun_out/r6p_inflate.log
timeout 600 python tools/inflate_var_run.py 4096:64 1024:64 256:64 >> gpurun_out/r6p_inflate.log 2>&1
cat gpurun_out/r6p_inflate.log
