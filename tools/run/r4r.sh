cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
for v in ov1 ov2 ov4 ov99; do
QATZIP_AMD_SO=build/var/lib_$v.so timeout 90 python tools/inflate_var_run.py 4096:64 2048:64 1024:64 1024:128 256:64 64:64 2>&1 | grep -v "^\["
done
