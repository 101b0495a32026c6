cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r4i.log
run() { # label envs...
  lab=$1; shift
  env "$@" timeout 90 python tools/inflate_var_run.py 16:64 64:64 256:64 512:64 1024:128 256:128 2048:64 2>&1 | grep -v "^\[" | sed "s/^default  */$lab  /" >> gpurun_out/r4i.log
}
run "wave     " QATZIP_AMD_INFLATE=wave
run "lane K=4 " QATZIP_AMD_INFLATE=lane QATZIP_AMD_INFLATE_K=4
run "lane K=8 " QATZIP_AMD_INFLATE=lane QATZIP_AMD_INFLATE_K=8
run "lane K=1 " QATZIP_AMD_INFLATE=lane QATZIP_AMD_INFLATE_K=1
sort -k3n -k6n -s gpurun_out/r4i.log
