#!/bin/bash
# rocprofv3 PMC passes over one command (each pass = its own run, kernel-trace only):
#   tools/pmc.sh <outdir> <cmd...>
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3"; do
  (cd $R && rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/$OUT/p$i -- "$@") > $R/$OUT/p$i.log 2>&1
  i=$((i+1))
done
