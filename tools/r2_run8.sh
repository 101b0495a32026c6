#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep "^{" | cut -c1-1400
