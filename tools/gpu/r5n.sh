# round 5, call n: FINAL sources -- whole GPU suite, smoke, config 5 (LoRA: fuse + rebuild cost), prefetch 1 vs 2 in-step (5 rounds)
mkdir -p gpurun_out/r5n
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x -s --durations=12 ) > gpurun_out/r5n/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5n/pytest_full.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r5n/smoke.log 2>&1
( time python bench.py --config 5 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline ) > gpurun_out/r5n/bench_config5.json 2> gpurun_out/r5n/bench_config5.err
timeout 600 python tools/ab_step.py --rounds 5 --variant prefetch1: --variant prefetch2:prefetch=2 > gpurun_out/r5n/ab_prefetch.log 2>&1
rm -rf gpurun_out/step_trace_config*/
grep -E "passed|failed|rc=" gpurun_out/r5n/pytest_full.log | tail -n 3; grep smoke gpurun_out/r5n/smoke.log; head -c 200 gpurun_out/r5n/bench_config5.json; echo; grep -o '"lora_fuse_s[^}]*' gpurun_out/r5n/bench_config5.json; grep -v amdgpu gpurun_out/r5n/ab_prefetch.log
