#!/bin/bash
# round 3, GPU batch C: real-geometry parity additions (selected by $1, default: the tiny harness twins)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
SEL=${1:-tiny}
timeout 3000 python -m pytest tests/test_full_geometry_gpu.py -x -q -s -k "$SEL" > gpurun_out/r3c/pytest_$2.log 2>&1
echo "rc=$?" >> gpurun_out/r3c/pytest_$2.log
grep -E "^\[|  ok |  BAD|  -- |passed|failed|rc=|Error|error" gpurun_out/r3c/pytest_$2.log | tail -70
