# round 6: K1 - a capped match (sixteen speculative bytes all equal) extended by its own lane, for all such lanes of the window at once (QZK_EXT_IT trips
# of four bytes; 0 = every capped lane through the serial exact path, as before)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6al_k1.log
for v in default ext0 ext6 ext24 default ext0 ext6 ext24; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6al_k1.log 2>&1
done
unset QATZIP_AMD_SO
cat gpurun_out/r6al_k1.log
timeout 300 python tools/prof_lz77.py silesia 16384 > gpurun_out/r6al_phases.txt 2>&1; cat gpurun_out/r6al_phases.txt
