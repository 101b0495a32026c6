#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for v in w16 w20 w22 w24 w24s; do echo "== $v"; QATZIP_AMD_SO=$R/build/var/lib_$v.so timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'], r['roofline']['chunks_per_launch'], r['roofline']['full_launch_alone_ms'], r['roofline']['other_kernels_ms'])"; done | tee gpurun_out/e_variants.log
timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/e_bench_full.log 2>&1; tail -c 3000 gpurun_out/e_bench_full.log
