#!/bin/bash
# usage: pmc_any.sh <tag> <python tool + args...>   -- runs five rocprofv3 --pmc passes of the command
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
run() { name=$1; shift; timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$name -- python $R/"${CMD[@]}" > $R/gpurun_out/$name.log 2>&1; grep -v "^[WE]2026" $R/gpurun_out/$name.log | tail -3; }
CMD=("$@")
run ${tag}A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
run ${tag}B SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU
run ${tag}C TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run ${tag}D TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run ${tag}E TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
