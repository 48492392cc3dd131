# round 5, call v: the new serving tests alone
mkdir -p gpurun_out/r5v
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "serving_sequence or thread_pool" --durations=5 ) > gpurun_out/r5v/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5v/pytest.log
tail -n 25 gpurun_out/r5v/pytest.log
