#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command (headline legs only)  -> gpurun_out/prof_stats
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes                  -> gpurun_out/prof_fetch, prof_write
#   3. --pmc TCC_HIT_sum TCC_MISS_sum                                            -> gpurun_out/prof_tcc
# then tools/pmc_summary.py turns 2. into profiles/<round>_pmc.json (run on the build box afterwards).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MB=${1:-4096}
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_tcc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --mb $MB --steps 2 --warmup 1 --no-cpu --no-extra > $R/gpurun_out/prof_stats.log 2>&1
grep '^{' $R/gpurun_out/prof_stats.log | tail -1 | cut -c1-300
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --mb $MB --steps 1 --warmup 1 --no-cpu --no-extra --no-probe > $R/gpurun_out/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --mb $MB --steps 1 --warmup 1 --no-cpu --no-extra --no-probe > $R/gpurun_out/prof_write.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/prof_tcc -- python $R/bench.py --mb $MB --steps 1 --warmup 1 --no-cpu --no-extra --no-probe > $R/gpurun_out/prof_tcc.log 2>&1
grep '^{' $R/gpurun_out/prof_tcc.log | tail -1 | cut -c1-200
