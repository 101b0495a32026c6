# round 6: K1's phase clocks on the round's final kernel (4 x 5 workgroups, slot tables 256 / 128 / 128, table rows of five workgroups)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python tools/prof_lz77.py silesia 16384 > gpurun_out/r6ak_phases.txt 2>&1
cat gpurun_out/r6ak_phases.txt
