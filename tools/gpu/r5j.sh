# round 5, call j: tile config 17 (192-row tiles of the one-wave-per-SIMD GEMM) -- parity, isolated timing, config 3 in-step A/B
mkdir -p gpurun_out/r5j
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -s -k "tile_config_17 or canary or production_shapes" ) > gpurun_out/r5j/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r5j/pytest.log
for shp in 2816,3072,15360 2816,3072,12288; do
  timeout 200 python tools/gemm_probe.py --shape $shp --ab 16,17 --epi gate --iters 12 --rotate 6 --pairs --check >> gpurun_out/r5j/probe.log 2>&1
done
timeout 200 python tools/gemm_probe.py --shape 2816,3072,12288 --groups 512,2304 --ab 16,17 --epi gate --iters 12 --rotate 6 --pairs --check >> gpurun_out/r5j/probe.log 2>&1
timeout 600 python tools/ab_step.py --height 768 --width 768 --embedders --rounds 3 --check --variant tile192: --variant tile256:gemm_tile192=0 > gpurun_out/r5j/ab_768.log 2>&1
timeout 600 python tools/ab_step.py --rounds 2 --check --variant tile192: --variant tile256:gemm_tile192=0 > gpurun_out/r5j/ab_1024.log 2>&1
( time python bench.py --config 3 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline ) > gpurun_out/r5j/bench_config3.json 2> gpurun_out/r5j/bench_config3.err
cp gpurun_out/step_trace_config3/steady_step.txt gpurun_out/r5j/steady_step_config3.txt 2>/dev/null
rm -rf gpurun_out/step_trace_config*/
tail -n 4 gpurun_out/r5j/pytest.log; grep -v amdgpu gpurun_out/r5j/probe.log | tail -n 12; grep -v amdgpu gpurun_out/r5j/ab_768.log gpurun_out/r5j/ab_1024.log; head -c 250 gpurun_out/r5j/bench_config3.json
