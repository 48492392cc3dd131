# round 5, call e: whole GPU suite (timing of the prefetched oracle), the driver's bench command with the in-step trace, preflight, configs 4 and 3
mkdir -p gpurun_out/r5e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x --durations=20 ) > gpurun_out/r5e/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5e/pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5e/bench_driver_cmd.json 2> gpurun_out/r5e/bench_driver_cmd.err
cp -r gpurun_out/step_trace_config2 gpurun_out/r5e/ 2>/dev/null
( time python bench.py --gpus 1 --preflight --single-rank-group ) > gpurun_out/r5e/preflight_single_rank_nccl.json 2> gpurun_out/r5e/preflight.err
( time python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/r5e/bench_config4.json 2> gpurun_out/r5e/bench_config4.err
cp -r gpurun_out/step_trace_config4 gpurun_out/r5e/ 2>/dev/null
( time python bench.py --config 3 --steps 20 --warmup 3 --no-pmc ) > gpurun_out/r5e/bench_config3.json 2> gpurun_out/r5e/bench_config3.err
cp -r gpurun_out/step_trace_config3 gpurun_out/r5e/ 2>/dev/null
tail -n 4 gpurun_out/r5e/pytest.log; head -c 600 gpurun_out/r5e/bench_driver_cmd.json; echo; cat gpurun_out/r5e/preflight_single_rank_nccl.json; head -c 400 gpurun_out/r5e/bench_config4.json; echo; head -c 400 gpurun_out/r5e/bench_config3.json
