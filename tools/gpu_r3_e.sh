#!/bin/bash
# round 3, GPU batch E: split-K for the small-M launches: op tests, config-1 bench (+ kernel trace), schnell full-depth parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -s -k "split_k" > gpurun_out/r3e/pytest_splitk.log 2>&1; echo "rc=$?" >> gpurun_out/r3e/pytest_splitk.log
grep -E "split-K|passed|failed|rc=|Error" gpurun_out/r3e/pytest_splitk.log | tail -12
FLUXMI_GEMM_SPLITK=0 timeout 300 python bench.py --config 1 --steps 16 --warmup 2 --no-cpu-baseline --no-pmc > gpurun_out/r3e/bench_c1_off.json 2> gpurun_out/r3e/bench_c1_off.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3e/prof -o c1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config 1 --steps 16 --warmup 2 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/r3e/bench_c1.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3e/bench_c1.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,json
for n in ("bench_c1_off","bench_c1"):
    try:
        d=json.loads(open(f'gpurun_out/r3e/{n}.json').read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(n, "failed", e)
for f in glob.glob('gpurun_out/r3e/prof/**/*kernel_stats.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'at::native' not in r['Name']]
    for r in rows[:12]:
        print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
if [ "$1" = "parity" ]; then
  timeout 1500 python -m pytest tests/test_full_geometry_gpu.py -x -q -s -k "schnell" > gpurun_out/r3e/pytest_schnell.log 2>&1; echo "rc=$?" >> gpurun_out/r3e/pytest_schnell.log
  grep -E "  ok |  BAD|passed|failed|rc=" gpurun_out/r3e/pytest_schnell.log | tail -8
fi
