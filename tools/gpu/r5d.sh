# round 5, call d: balanced attention grid -- parity, isolated timing, in-step A/B, PMC
mkdir -p gpurun_out/r5d
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -s -k "attention" ) > gpurun_out/r5d/pytest_attn.log 2>&1
echo "rc=$?" >> gpurun_out/r5d/pytest_attn.log
( time timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_full_geometry_gpu.py -m gpu -q -x -s -k "fused_equals or c2_loop4 or tiny or c3_2p2 or batch_sharded" ) > gpurun_out/r5d/pytest_engine.log 2>&1
echo "rc=$?" >> gpurun_out/r5d/pytest_engine.log
timeout 300 python tools/attn_probe.py --split 0,1 > gpurun_out/r5d/attn_probe.log 2>&1
timeout 300 python tools/attn_probe.py --split 0,1 --L 2816 >> gpurun_out/r5d/attn_probe.log 2>&1
timeout 300 python tools/attn_probe.py --split 0,1 --B 2 >> gpurun_out/r5d/attn_probe.log 2>&1
timeout 600 python tools/ab_step.py --variant split:attn_split=1 --variant split_pf:attn_split=2 --variant nosplit:attn_split=0 --rounds 3 --check > gpurun_out/r5d/ab_step.log 2>&1
timeout 600 python tools/ab_step.py --height 768 --width 768 --embedders --variant split:attn_split=1 --variant nosplit:attn_split=0 --rounds 3 > gpurun_out/r5d/ab_step_768.log 2>&1
cd /tmp
for i in 1 2; do
  case $i in
   1) C="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES";;
   2) C="GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA";;
  esac
  timeout 120 rocprofv3 --kernel-trace --pmc $C -d $GRAFT_REPO_ROOT/gpurun_out/r5d/pmc_attn/p$i -o runc --output-format csv -- python $GRAFT_REPO_ROOT/tools/attn_probe.py --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/r5d/pmc_attn_p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/r5d/pmc_attn -k attention > gpurun_out/r5d/pmc_attn.txt 2>&1
tail -n 3 gpurun_out/r5d/pytest_attn.log gpurun_out/r5d/pytest_engine.log; cat gpurun_out/r5d/attn_probe.log gpurun_out/r5d/ab_step.log gpurun_out/r5d/ab_step_768.log
