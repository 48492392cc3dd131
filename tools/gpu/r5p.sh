# round 5, call p: counters on the FINAL sources (the committed PMC files are keyed by a hash of the kernel sources)
mkdir -p gpurun_out/r5p
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-step ) > gpurun_out/r5p/bench_driver_cmd.json 2> gpurun_out/r5p/bench.err
cp gpurun_out/step_trace_config2/steady_step.txt gpurun_out/r5p/steady_step_config2.txt 2>/dev/null
( time python bench.py --config 3 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --pmc-step ) > gpurun_out/r5p/bench_config3.json 2> gpurun_out/r5p/bench3.err
cp gpurun_out/step_trace_config3/steady_step.txt gpurun_out/r5p/steady_step_config3.txt 2>/dev/null
cp profiles/r05_mfma_config2.json profiles/r05_traffic_config2.json profiles/r05_step_pmc_config2.json profiles/r05_step_pmc_config3.json gpurun_out/r5p/ 2>/dev/null
rm -rf gpurun_out/step_trace_config*/ gpurun_out/pmc_config*/ gpurun_out/step_pmc_config*/
head -c 200 gpurun_out/r5p/bench_driver_cmd.json; echo; head -c 200 gpurun_out/r5p/bench_config3.json; echo; ls gpurun_out/r5p
