# round 5, fourteenth GPU call: the LZ4 compressor with the collision-free insert path alone (l4f) against the round's build (default)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r5n_lz4.log
for v in default l4f default l4f; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/legs_run.py lz4 1024 >> gpurun_out/r5n_lz4.log 2>&1
done
unset QATZIP_AMD_SO
cut -c1-200 gpurun_out/r5n_lz4.log
