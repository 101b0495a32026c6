# round 6: phase A, the second half of the first round of waves started late (QATZIP_AMD_INFLATE_STAGGER=ticks:lo:hi)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6t_inflate.log
for s in 0:0:0 50000:1024:2048 100000:1024:2048 150000:1024:2048 100000:0:1024 0:0:0 100000:1024:2048; do
  echo "== stagger $s" >> gpurun_out/r6t_inflate.log
  QATZIP_AMD_INFLATE_STAGGER=$s timeout 600 python tools/inflate_var_run.py 4096:64 >> gpurun_out/r6t_inflate.log 2>&1
done
# every other group of 128 workgroups
cat gpurun_out/r6t_inflate.log
