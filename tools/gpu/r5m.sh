# round 5, call m: tile config 17 with bf16 operands (config 1 / text encoders)
mkdir -p gpurun_out/r5m
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_text_gpu.py tests/test_full_geometry_gpu.py -m gpu -q -x -s -k "tile_config_17 or split_k or text or schnell or bf16_gemm" ) > gpurun_out/r5m/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r5m/pytest.log
FLUXMI_GEMM_TILE192=0 python bench.py --config 1 --steps 28 --warmup 3 --no-pmc --no-cpu-baseline > gpurun_out/r5m/bench_config1_tile256.json 2> gpurun_out/r5m/b1.err
cp gpurun_out/step_trace_config1/steady_step.txt gpurun_out/r5m/steady_step_config1_tile256.txt 2>/dev/null
FLUXMI_GEMM_TILE192=1 python bench.py --config 1 --steps 28 --warmup 3 --no-pmc > gpurun_out/r5m/bench_config1_tile192.json 2> gpurun_out/r5m/b2.err
cp gpurun_out/step_trace_config1/steady_step.txt gpurun_out/r5m/steady_step_config1_tile192.txt 2>/dev/null
FLUXMI_GEMM_TILE192=0 python bench.py --config 1 --steps 28 --warmup 3 --no-pmc --no-cpu-baseline --no-step-trace > gpurun_out/r5m/bench_config1_tile256_b.json 2> gpurun_out/r5m/b3.err
python tools/text_probe.py > gpurun_out/r5m/text_probe.log 2>&1
rm -rf gpurun_out/step_trace_config*/
tail -n 4 gpurun_out/r5m/pytest.log; for f in gpurun_out/r5m/bench_config1_tile256.json gpurun_out/r5m/bench_config1_tile192.json gpurun_out/r5m/bench_config1_tile256_b.json; do head -c 160 $f; echo; done; grep -v amdgpu gpurun_out/r5m/text_probe.log | tail -n 6
