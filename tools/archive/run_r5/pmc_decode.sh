cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for g in "A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
         "B SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU" \
         "D TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "E TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE"; do
  set -- $g; name=dec$1; shift
  rm -rf $R/gpurun_out/$name
  timeout 100 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$name -- python $R/tools/inflate_var_run.py 4096:64 > $R/gpurun_out/$name.log 2>&1
  grep "inflate" $R/gpurun_out/$name.log | tail -1 | cut -c1-160
done
