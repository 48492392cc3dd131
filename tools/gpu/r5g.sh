# round 5, call g: the whole GPU suite on the final sources (log kept under profiles/), smoke, the driver's bench command, config 1 with its step trace,
# and the in-step contribution of every kernel-selection feature (tools/ab_step.py)
mkdir -p gpurun_out/r5g
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x -s --durations=15 ) > gpurun_out/r5g/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5g/pytest_full.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r5g/smoke.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5g/bench_driver_cmd.json 2> gpurun_out/r5g/bench_driver_cmd.err
cp gpurun_out/step_trace_config2/steady_step.txt gpurun_out/r5g/steady_step_config2.txt 2>/dev/null
( time python bench.py --config 1 --steps 28 --warmup 3 --no-pmc ) > gpurun_out/r5g/bench_config1.json 2> gpurun_out/r5g/bench_config1.err
cp gpurun_out/step_trace_config1/steady_step.txt gpurun_out/r5g/steady_step_config1.txt 2>/dev/null
timeout 900 python tools/ab_step.py --rounds 3 --check --variant default: --variant no_prefetch:prefetch=0 --variant prefetch2:prefetch=2 --variant no_w_pairs:w_pairs=0 --variant no_persist:gemm_persist=0,fuse_kv=1 --variant relayout_k:fuse_kv=1 --variant no_qlut:qlut=0 --variant attn_split_forced:attn_split=2 > gpurun_out/r5g/ab_features.log 2>&1
rm -rf gpurun_out/step_trace_config*/ gpurun_out/pmc_config*/
du -sh gpurun_out
grep -E "passed|failed|rc=" gpurun_out/r5g/pytest_full.log | tail -n 3; tail -n 2 gpurun_out/r5g/smoke.log; head -c 250 gpurun_out/r5g/bench_driver_cmd.json; echo; cat gpurun_out/r5g/ab_features.log | grep -v amdgpu
