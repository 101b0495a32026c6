# one gpurun call at the end of a round: the profile passes, the decode kernels' counters, phase A's per-segment times, the
# bench lines, the fuzz campaigns on the hardware, the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
bash tools/profile_round.sh 4096 > gpurun_out/profile_round.log 2>&1
bash tools/run/pmc_decode.sh > gpurun_out/pmc_decode.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_show.py dec qzk_inflate_spec qzk_lz_resolve > gpurun_out/decode_counters.txt 2>&1
QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so timeout 300 python tools/prof_phaseA_tail.py 64 256 1024 4096 > gpurun_out/phaseA_tail.txt 2>&1
for mb in 64 4096; do QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so timeout 120 python tools/prof_spec.py $mb 64 2>/dev/null | head -16; done > gpurun_out/phaseA_waves.txt
bash tools/run/bench_final.sh > gpurun_out/bench_final.log 2>&1
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd bash tools/small_calls.sh > gpurun_out/small_calls.txt 2>&1
API_PASSES=5 timeout 200 python tools/api_h2h.py 2047 default 2 > gpurun_out/api_h2h.txt 2>&1
timeout 200 python tools/inflate_var_run.py 64:64 256:64 1024:64 2048:64 4096:64 1024:128 1024:16 1024:256 1024:512 > gpurun_out/inflate_sizes.txt 2>&1
bash tools/run/fuzz_gpu.sh > gpurun_out/fuzz_both.log 2>&1
bash tools/run/full_gpu.sh > gpurun_out/full_gpu_tail.log 2>&1
tail -n 5 gpurun_out/full_gpu_tail.log; tail -n 12 gpurun_out/phaseA_tail.txt; tail -n 30 gpurun_out/bench_final.log | cut -c1-900
