# round 5, call l: final sources -- whole GPU suite (log -> profiles/), smoke, driver's bench command, config 3
mkdir -p gpurun_out/r5l
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x -s --durations=15 ) > gpurun_out/r5l/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5l/pytest_full.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r5l/smoke.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5l/bench_driver_cmd.json 2> gpurun_out/r5l/bench_driver_cmd.err
cp gpurun_out/step_trace_config2/steady_step.txt gpurun_out/r5l/steady_step_config2.txt 2>/dev/null
cp profiles/r05_mfma_config2.json profiles/r05_traffic_config2.json gpurun_out/r5l/ 2>/dev/null
( time python bench.py --config 3 --steps 20 --warmup 3 --no-pmc ) > gpurun_out/r5l/bench_config3.json 2> gpurun_out/r5l/bench_config3.err
cp gpurun_out/step_trace_config3/steady_step.txt gpurun_out/r5l/steady_step_config3.txt 2>/dev/null
timeout 300 python tools/ab_step.py --height 512 --width 512 --rounds 2 --variant default: --variant nosplit:attn_split=0 > gpurun_out/r5l/ab_512.log 2>&1
rm -rf gpurun_out/step_trace_config*/ gpurun_out/pmc_config*/
grep -E "passed|failed|rc=" gpurun_out/r5l/pytest_full.log | tail -n 3; grep smoke gpurun_out/r5l/smoke.log; head -c 250 gpurun_out/r5l/bench_driver_cmd.json; echo; head -c 250 gpurun_out/r5l/bench_config3.json; echo; grep -v amdgpu gpurun_out/r5l/ab_512.log
