#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_golden.py -x -q -s 2>&1 | tail -15 | tee gpurun_out/i_pytest.log
QATZIP_AMD_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --mb 1024 --steps 1 --warmup 1 2>&1 | grep -v Warning | tail -5 | cut -c1-2500 | tee gpurun_out/i_bench2.log
