cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
for k in 4 1; do
  export QATZIP_AMD_INFLATE_K=$k
  timeout 300 bash tools/pmc_any.sh pb$k tools/inflate_var_run.py 4096:64 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_show.py pb$k inflate_spec inflate_tok > gpurun_out/r4g_pmc_k$k.txt 2>&1
done
cat gpurun_out/r4g_pmc_k4.txt; grep "tok_kernel" gpurun_out/r4g_pmc_k1.txt
