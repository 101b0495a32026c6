cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_xnolit.so
# counters of phase A at full occupancy (4 GiB), 4 lanes per segment against 1; the literal stores compiled out so that no
# segment is handed back (time only)
for k in 4 1; do
  export QATZIP_AMD_INFLATE_K=$k
  timeout 400 bash tools/pmc_any.sh pa$k tools/inflate_var_run.py 4096:64 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_show.py pa$k inflate_spec inflate_tok > gpurun_out/r4e_pmc_k$k.txt 2>&1
done
cat gpurun_out/r4e_pmc_k4.txt gpurun_out/r4e_pmc_k1.txt
