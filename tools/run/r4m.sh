cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
for k in 4 1; do
echo "== K=$k"
QATZIP_AMD_INFLATE_K=$k QATZIP_AMD_TRACE=1 timeout 90 python tools/inflate_var_run.py 4096:64 2>&1 | tail -n 12
done
