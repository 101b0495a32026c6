#!/bin/bash
# round-2 K1 measurement on the GPU box: parity of the compress kernels, a short bench, kernel stats, fabric traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
cd $R
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/${TAG}_bench.log 2>&1; grep '^{' gpurun_out/${TAG}_bench.log | tail -1 | cut -c1-1500
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/${TAG}_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu > $R/gpurun_out/${TAG}_write.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_tcc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu > $R/gpurun_out/${TAG}_tcc.log 2>&1
ls $R/gpurun_out/${TAG}_stats/*/ | head
