#!/bin/bash
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
i=1
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  (cd $R && timeout 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/$OUT/p$i -- python tools/attn_probe.py --iters 2) > $R/$OUT/p$i.log 2>&1
  i=$((i+1))
done
cd $R; python tools/pmc_summary.py $OUT -k attention; python - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/p1/*/*kernel_trace.csv')[0]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in csv.DictReader(open(f)) if 'attention' in r['Kernel_Name']]
print('durations us', d[-6:])
PY
