# round 6: K1 on launches of a few chunks per wave - fewer resident waves so that the rounds come out even (QATZIP_AMD_K1_WGS = waves in all)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6ah_k1.log
for mb in 1024 2048 512; do
for w in 5120 4096 3072 5120 4096; do
  echo "== $mb MiB, $w waves" >> gpurun_out/r6ah_k1.log
  QATZIP_AMD_K1_WGS=$w timeout 300 python tools/k1_var_run.py $mb 2>&1 | grep compress >> gpurun_out/r6ah_k1.log
done; done
cat gpurun_out/r6ah_k1.log
