# round 5, call x: the two ragged real-width parity cases (reference's default 720 x 1024, and an odd L = 3257 on the balanced attention grid)
mkdir -p gpurun_out/r5x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -x -s -k "default_1p1 or ragged_1p1" ) > gpurun_out/r5x/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5x/pytest.log
grep -c "  ok " gpurun_out/r5x/pytest.log; grep -n "BAD\|passed\|failed\|rc=\|Error" gpurun_out/r5x/pytest.log | head -40
