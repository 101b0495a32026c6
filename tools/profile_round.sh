#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command (headline legs only)  -> gpurun_out/prof_stats
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes                  -> gpurun_out/prof_fetch, prof_write
#   3. --pmc TCC_HIT_sum TCC_MISS_sum                                            -> gpurun_out/prof_tcc
#   4. the same three for BASELINE configs 3 and 4 alone (tools/legs_run.py lz4 / raw16 / raw64 / raw128, 1 GiB)
#                                                                               -> gpurun_out/prof_<leg>_{stats,fetch,write}
# then tools/pmc_summary.py turns the --pmc passes into profiles/<round>_*_pmc.json (run on the build box afterwards, before
# the sources change: the summaries carry their hash).  usage: profile_round.sh [MiB of the bench] [legs: "lz4 raw16 raw64 raw128"]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
MB=${1:-4096}
LEGS=${2:-"lz4 raw16 raw64 raw128"}
prof() { # dir, rocprof flags..., -- command
  d=$1; shift
  rm -rf $R/gpurun_out/$d
  timeout 300 rocprofv3 "$@" > $R/gpurun_out/$d.log 2>&1
}
B="python $R/bench.py --mb $MB --steps 1 --warmup 1 --no-cpu --no-extra --no-probe"
prof prof_stats --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --mb $MB --steps 2 --warmup 1 --no-cpu --no-extra
grep '^{' $R/gpurun_out/prof_stats.log | tail -1 | cut -c1-300
prof prof_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- $B
prof prof_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- $B
prof prof_tcc --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/prof_tcc -- $B
for leg in $LEGS; do
  L="python $R/tools/legs_run.py $leg 1024"
  prof prof_${leg}_stats --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${leg}_stats -- $L
  grep '^{' $R/gpurun_out/prof_${leg}_stats.log | tail -1 | cut -c1-400
  prof prof_${leg}_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${leg}_fetch -- $L
  prof prof_${leg}_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${leg}_write -- $L
done
ls $R/gpurun_out | grep "^prof_" | tr '\n' ' '
