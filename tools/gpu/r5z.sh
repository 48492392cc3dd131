# round 5, call z: stem linears on one tile config (batch-invariant fp8 flow): the B = 32 test + the invariance probe first; only if green, the
# driver's bench command with the in-step counters on the new sources, then the whole GPU suite + smoke
mkdir -p gpurun_out/r5z
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 200 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "maximum_batch or large_batches or serving_sequence" ) > gpurun_out/r5z/pytest_first.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r5z/pytest_first.log
tail -n 4 gpurun_out/r5z/pytest_first.log
( timeout 120 python tools/probes/batch_invariance_probe.py --B 2,4,8,32 ) > gpurun_out/r5z/invariance.log 2>&1
grep "^B =" gpurun_out/r5z/invariance.log
if [ $rc -ne 0 ]; then exit 1; fi
( time python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-step ) > gpurun_out/r5z/bench_driver_cmd.json 2> gpurun_out/r5z/bench.err
cp gpurun_out/step_trace_config2/steady_step.txt gpurun_out/r5z/steady_step_config2.txt 2>/dev/null
cp profiles/r05_step_pmc_config2.json gpurun_out/r5z/ 2>/dev/null
rm -rf gpurun_out/step_trace_config*/ gpurun_out/pmc_config*/ gpurun_out/step_pmc_config*/
head -c 250 gpurun_out/r5z/bench_driver_cmd.json; echo
( time python -m pytest tests -m gpu -q -x -s --durations=12 ) > gpurun_out/r5z/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5z/pytest_full.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r5z/smoke.log 2>&1
grep -E "passed|failed|rc=" gpurun_out/r5z/pytest_full.log | tail -n 3; grep smoke gpurun_out/r5z/smoke.log
