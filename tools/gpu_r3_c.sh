#!/bin/bash
# round 3, GPU batch C: new host-surface tests (reference configs, engine batch > 8) + engine timing window
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/r3c; mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "reference_config or pipeline_generate_latents or prequantized or denoise_loop or batch_sharded" > gpurun_out/r3c/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3c/pytest.log
tail -15 gpurun_out/r3c/pytest.log
