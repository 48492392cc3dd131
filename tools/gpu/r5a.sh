mkdir -p gpurun_out/r5a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x --durations=60 ) > gpurun_out/r5a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5a/pytest.log
# PMC of the shipped attention kernel (3 passes)
cd /tmp
for i in 1 2 3; do
  case $i in
   1) C="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES";;
   2) C="GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA";;
   3) C="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SALU";;
  esac
  timeout 120 rocprofv3 --kernel-trace --pmc $C -d $GRAFT_REPO_ROOT/gpurun_out/r5a/pmc_attn/p$i -o runc --output-format csv -- python $GRAFT_REPO_ROOT/tools/attn_probe.py --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/r5a/pmc_attn_p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/r5a/pmc_attn -k attention > gpurun_out/r5a/pmc_attn.txt 2>&1
tail -5 gpurun_out/r5a/pytest.log
