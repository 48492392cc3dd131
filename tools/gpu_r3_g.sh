#!/bin/bash
# round 3, GPU batch G: the attention tests, the engine tests and the real-geometry parity tests with the 4-wave attention kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_full_geometry_gpu.py -x -q -m gpu -k "attention or engine or geometry or real or schnell or loop or lora or batch or denoise or config" -s ) > gpurun_out/r3g/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3g/pytest.log
grep -E "passed|failed|error|rc=|4-wave|byte|identical|rel-L2" gpurun_out/r3g/pytest.log | tail -40
