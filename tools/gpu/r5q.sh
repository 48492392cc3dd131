# round 5, call q: the N > 1 code path of bench.py over a ONE-rank RCCL group on the GPU box (broadcast, in-step amax all-reduce, barriers, rank-id check)
mkdir -p gpurun_out/r5q
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python bench.py --gpus 1 --single-rank-group --steps 20 --warmup 5 --no-pmc --no-cpu-baseline ) > gpurun_out/r5q/bench_single_rank_group_nccl.json 2> gpurun_out/r5q/bench.err
( time python bench.py --gpus 1 --single-rank-group --config 4 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-step-trace ) > gpurun_out/r5q/bench_config4_single_rank_group.json 2> gpurun_out/r5q/bench4.err
rm -rf gpurun_out/step_trace_config*/
wc -l gpurun_out/r5q/*.json; head -c 300 gpurun_out/r5q/bench_single_rank_group_nccl.json; echo; grep -o '"nranks[^,]*,[^,]*,[^,]*' gpurun_out/r5q/*.json; tail -n 3 gpurun_out/r5q/bench.err gpurun_out/r5q/bench4.err
