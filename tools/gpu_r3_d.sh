#!/bin/bash
# round 3, GPU batch D: op / engine / text suites after the GEMM clean-up, then the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_text_gpu.py -x -q > gpurun_out/r3d/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3d/pytest.log
tail -6 gpurun_out/r3d/pytest.log
if [ "$1" = "bench" ]; then
  timeout 1200 python bench.py > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
  tail -c 3000 gpurun_out/r3d/bench.json
  tail -5 gpurun_out/r3d/bench.err
fi
