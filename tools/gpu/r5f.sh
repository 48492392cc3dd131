# round 5, call f: bench lines (driver's command, configs 4 / 3 / 5 / 1), preflight, and the gemm_w1 "all 256 CUs busy" probe
mkdir -p gpurun_out/r5f
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5f/bench_driver_cmd.json 2> gpurun_out/r5f/bench_driver_cmd.err
cp gpurun_out/step_trace_config2/steady_step.txt gpurun_out/r5f/steady_step_config2.txt 2>/dev/null
cp profiles/r04_mfma_config2.json profiles/r04_traffic_config2.json gpurun_out/r5f/ 2>/dev/null
( time python bench.py --gpus 1 --preflight --single-rank-group ) > gpurun_out/r5f/preflight_single_rank_nccl.json 2> gpurun_out/r5f/preflight.err
( time python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/r5f/bench_config4.json 2> gpurun_out/r5f/bench_config4.err
cp gpurun_out/step_trace_config4/steady_step.txt gpurun_out/r5f/steady_step_config4.txt 2>/dev/null
( time python bench.py --config 3 --steps 20 --warmup 3 --no-pmc ) > gpurun_out/r5f/bench_config3.json 2> gpurun_out/r5f/bench_config3.err
cp gpurun_out/step_trace_config3/steady_step.txt gpurun_out/r5f/steady_step_config3.txt 2>/dev/null
( time python bench.py --config 5 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline ) > gpurun_out/r5f/bench_config5.json 2> gpurun_out/r5f/bench_config5.err
( time python bench.py --config 1 --steps 28 --warmup 3 --no-pmc ) > gpurun_out/r5f/bench_config1.json 2> gpurun_out/r5f/bench_config1.err
cp gpurun_out/step_trace_config1/steady_step.txt gpurun_out/r5f/steady_step_config1.txt 2>/dev/null
# gemm_w1 (tile config 16, K = 15360, gate*y+x epilogue, weights rotated through 6 copies = HBM-cold like in the step): launches of 128 / 192 / 216 / 256 tiles
for shp in 2048,4096,15360 4096,3072,15360 4608,3072,15360 4096,4096,15360 4608,3072,12288 4096,4096,12288; do
  timeout 200 python tools/gemm_probe.py --shape $shp --cfg 16 --epi gate --iters 12 --rotate 6 --pairs >> gpurun_out/r5f/w1_fit_probe.log 2>&1
done
rm -rf gpurun_out/step_trace_config*/
du -sh gpurun_out
head -c 300 gpurun_out/r5f/bench_driver_cmd.json; echo; tail -n 8 gpurun_out/r5f/w1_fit_probe.log
