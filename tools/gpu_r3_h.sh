#!/bin/bash
# round 3, GPU batch H (final evidence): the default bench run (config 2, live PMC, CPU baseline) exactly as the driver calls it, a
# rocprofv3 --kernel-trace --stats run of a short bench (steady-step summary), and the bench lines of configs 1, 3, 5
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > gpurun_out/r3h/bench_c2.json 2> gpurun_out/r3h/bench_c2.err
tail -c 600 gpurun_out/r3h/bench_c2.err | tail -3
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3h/prof -o c2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/r3h/bench_c2_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3h/bench_c2_prof.err
cd $GRAFT_REPO_ROOT
for c in 1 3 5; do
  timeout 400 python bench.py --config $c --no-pmc --no-cpu-baseline > gpurun_out/r3h/bench_c$c.json 2> gpurun_out/r3h/bench_c$c.err
done
python - <<'PY'
import json
for n in ("bench_c2","bench_c2_prof","bench_c1","bench_c3","bench_c5"):
    try:
        d=json.loads(open(f'gpurun_out/r3h/{n}.json').read().strip().splitlines()[-1]); print(n, d["value"], d["unit"], d["ms_per_step"], 'roofline', d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("attention",{}).get("frac"))
    except Exception as e: print(n, "failed", e)
PY
ls gpurun_out/r3h/prof/*/ 2>/dev/null | head
