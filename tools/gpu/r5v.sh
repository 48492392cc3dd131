# round 5, call v: B = 32 at real width
mkdir -p gpurun_out/r5v
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "maximum_batch" --durations=5 ) > gpurun_out/r5v/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5v/pytest.log
tail -n 25 gpurun_out/r5v/pytest.log
