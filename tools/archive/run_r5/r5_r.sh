# round 5, eighteenth GPU call: what the looks at trails cost a wave of phase A (-DQZK_SPEC_PROF), 64 MiB / 1 GiB / 4 GiB
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so
for mb in 64 1024 4096; do timeout 300 python tools/prof_spec.py $mb 64 2>&1 | head -22; done > gpurun_out/r5r_looks.txt
cut -c1-200 gpurun_out/r5r_looks.txt
