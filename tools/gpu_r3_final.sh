#!/bin/bash
# round 3, final GPU batch: the default bench run exactly as the driver calls it (config 2, live PMC, CPU baseline), a rocprofv3
# --kernel-trace --stats run of a short bench (steady-step summary), the bench lines of configs 1, 3, 5, then the whole GPU suite + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3z
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > gpurun_out/r3z/bench_c2.json 2> gpurun_out/r3z/bench_c2.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3z/prof -o c2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/r3z/bench_c2_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3z/bench_c2_prof.err
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r3z/prof/*agent_info* gpurun_out/r3z/prof/*domain_stats*
for c in 1 3 5; do
  timeout 400 python bench.py --config $c --no-pmc > gpurun_out/r3z/bench_c$c.json 2> gpurun_out/r3z/bench_c$c.err
done
python - <<'PY'
import json
for n in ("bench_c2","bench_c2_prof","bench_c1","bench_c3","bench_c5"):
    try:
        d=json.loads(open(f'gpurun_out/r3z/{n}.json').read().strip().splitlines()[-1]); print(n, d["value"], d["unit"], d["ms_per_step"], 'roofline', d["roofline"]["frac"], d["roofline"].get("traffic"), 'cpu', (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(n, "failed", e)
PY
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r3z/pytest_gpu.log 2>&1
echo "rc=$?" >> gpurun_out/r3z/pytest_gpu.log
tail -8 gpurun_out/r3z/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
