# round 5, call i: hardware counters of the kernels inside the step (bench.py --pmc-step), configs 2 and 3
mkdir -p gpurun_out/r5i
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --pmc-step ) > gpurun_out/r5i/bench_pmc_step.json 2> gpurun_out/r5i/bench_pmc_step.err
cp profiles/r05_step_pmc_config2.json gpurun_out/r5i/ 2>/dev/null
ls -la gpurun_out/step_pmc_config2/*/ 2>/dev/null | head -20
for p in 0 1 2; do f=$(ls gpurun_out/step_pmc_config2/p$p/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -2 "$f" | cut -c1-300; wc -l "$f"); done > gpurun_out/r5i/csv_heads.txt 2>&1
rm -rf gpurun_out/step_trace_config*/ gpurun_out/pmc_config*/ gpurun_out/step_pmc_config*/
cat gpurun_out/r5i/r05_step_pmc_config2.json; tail -n 5 gpurun_out/r5i/bench_pmc_step.err; head -c 300 gpurun_out/r5i/bench_pmc_step.json
