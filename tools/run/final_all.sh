# one gpurun call at the end of a round: the profile passes, the bench lines, the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
bash tools/profile_round.sh 4096 > gpurun_out/profile_round.log 2>&1
bash tools/run/bench_final.sh > gpurun_out/bench_final.log 2>&1
bash tools/run/full_gpu.sh > gpurun_out/full_gpu_tail.log 2>&1
tail -n 5 gpurun_out/full_gpu_tail.log
