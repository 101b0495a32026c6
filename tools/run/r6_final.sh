# one gpurun call on the round's code: the profile passes (kernel stats + PMC of the bench command and of configs 3 / 4), the decode kernels'
# counters, the bench lines (default, two ranks, the eight-rank rehearsal on the one device), small calls, decode sizes, the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
bash tools/profile_round.sh 4096 > gpurun_out/profile_round.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in "A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
         "B SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU" \
         "D TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "E TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE"; do
  set -- $g; name=dec$1; shift
  rm -rf $R/gpurun_out/$name
  timeout 100 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$name -- python $R/tools/inflate_var_run.py 4096:64 > $R/gpurun_out/$name.log 2>&1
  grep "inflate" $R/gpurun_out/$name.log | tail -1 | cut -c1-160
done
cd $GRAFT_REPO_ROOT
python tools/pmc_show.py dec qzk_inflate_spec qzk_lz_resolve > gpurun_out/decode_counters.txt 2>&1
timeout 900 python bench.py > gpurun_out/bench_r6_final.json 2> gpurun_out/bench_r6_final.err; echo "bench rc $?"
timeout 400 python bench.py --gpus 2 --steps 1 --warmup 1 --mb 2048 --no-cpu > gpurun_out/bench_r6_2ranks.json 2> gpurun_out/bench_r6_2ranks.err; echo "bench2 rc $?"
export QATZIP_AMD_RCCL_TIMEOUT=5 QATZIP_AMD_BENCH_LEG_TIMEOUT=200
timeout 600 python bench.py --gpus 8 --mb 256 --members 16 --steps 1 --no-cpu > gpurun_out/bench_r6_8ranks.json 2> gpurun_out/bench_r6_8ranks.err; echo "bench8 rc $?"
unset QATZIP_AMD_RCCL_TIMEOUT QATZIP_AMD_BENCH_LEG_TIMEOUT
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd timeout 600 bash tools/small_calls.sh > gpurun_out/small_calls.txt 2>&1
timeout 300 python tools/inflate_var_run.py 64:64 256:64 1024:64 2048:64 4096:64 1024:128 1024:16 1024:256 1024:512 > gpurun_out/inflate_sizes.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu.log 2>&1; echo "rc $?" >> gpurun_out/full_gpu.log
tail -n 4 gpurun_out/full_gpu.log; tail -3 gpurun_out/profile_round.log | cut -c1-300
python - <<'PY'
import json
for f in ("bench_r6_final", "bench_r6_2ranks", "bench_r6_8ranks"):
    try:
        j = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, {k: j.get(k) for k in ("value", "n_gpus", "ms_per_step", "per_rank", "one_stream_gather_share", "one_stream_GBps")}, {k: j["config"].get(k) for k in ("compress_GBps", "decompress_GBps")})
    except Exception as e:
        print(f, "no line", e)
PY
