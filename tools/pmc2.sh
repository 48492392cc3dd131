#!/bin/bash
# memory-path PMC passes: tools/pmc2.sh <outdir> <cmd...>
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
i=1
for P in "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
         "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE" \
         "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  (cd $R && rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/$OUT/p$i -- "$@") > $R/$OUT/p$i.log 2>&1
  i=$((i+1))
done
cd $R; python tools/pmc_summary.py $OUT -k gemm
