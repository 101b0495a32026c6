# round 6: phase A - the host's launch order (segments by compressed length in classes of 2^s bytes, longest first; s = 5 is the product's): wider
# classes keep more stream neighbours together; and the waves' timeline with the order taken into account
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6af_inflate.log
for s in 5 8 10 12 16 5 8 10 12 16; do
  echo "== class shift $s" >> gpurun_out/r6af_inflate.log
  QATZIP_AMD_INFLATE_CLS=$s timeout 600 python tools/inflate_var_run.py 4096:64 1024:64 >> gpurun_out/r6af_inflate.log 2>&1
done
cat gpurun_out/r6af_inflate.log
for s in 5 16; do
  echo "== timeline, class shift $s"
  QATZIP_AMD_INFLATE_CLS=$s QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so timeout 600 python tools/prof_phaseA_timeline.py 4096
done > gpurun_out/r6af_timeline.log 2>&1
cat gpurun_out/r6af_timeline.log
