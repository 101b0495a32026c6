# round 5, thirty-second GPU call: small synchronous calls with the own-queue streams made only by a piece-wise decode
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd
bash tools/small_calls.sh > gpurun_out/r5af_small.txt 2>&1
cat gpurun_out/r5af_small.txt
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 default 2>&1 | cut -c1-250
