# round 5, call x: batch of two at an odd L, and the bf16 flow at large M
mkdir -p gpurun_out/r5x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -x -s -k "B2_ragged or bf16_1p1_L4352" ) > gpurun_out/r5x/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5x/pytest.log
grep -c "  ok " gpurun_out/r5x/pytest.log; grep -n "BAD\|passed\|failed\|rc=\|Error" gpurun_out/r5x/pytest.log | head -40
