# round 5, thirtieth GPU call: 256 KB and 512 KB segments with sixteen and thirty-two lanes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
for k in 16 32; do QATZIP_AMD_INFLATE_K=$k timeout 300 python tools/inflate_var_run.py 1024:512 1024:256 256:512 2>&1 | cut -c1-150; done > gpurun_out/r5ad.log
cat gpurun_out/r5ad.log
