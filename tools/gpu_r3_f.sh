#!/bin/bash
# round 3, GPU batch F: the whole GPU suite (what the driver runs at round end) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r3f/pytest_gpu.log 2>&1
echo "rc=$?" >> gpurun_out/r3f/pytest_gpu.log
tail -12 gpurun_out/r3f/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
