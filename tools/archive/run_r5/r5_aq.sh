# round 5, the last GPU seconds: the decode tests on the phase-B fix, then the PMC traffic passes again (their summaries are bound
# to the sources), the most important first
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; mkdir -p $R/gpurun_out
(cd $R && timeout 60 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -n 2) > $R/gpurun_out/aq_tests.log; cat $R/gpurun_out/aq_tests.log
prof() { d=$1; shift; rm -rf $R/gpurun_out/$d; timeout 60 rocprofv3 "$@" > $R/gpurun_out/$d.log 2>&1; echo "$d done"; }
B="python $R/bench.py --mb 4096 --steps 1 --warmup 1 --no-cpu --no-extra --no-probe"
prof prof_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- $B
prof prof_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- $B
for leg in raw64 lz4 raw16 raw128; do
  L="python $R/tools/legs_run.py $leg 1024"
  prof prof_${leg}_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${leg}_fetch -- $L
  prof prof_${leg}_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${leg}_write -- $L
done
