# round 5, call w: every BASELINE config on the FINAL sources, one lease (lines without counters / CPU leg; the headline line with both is box 7)
mkdir -p gpurun_out/r5w
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in 2 1 3 4 5; do
  case $c in 1) st="--steps 28 --warmup 3";; 2) st="--steps 20 --warmup 5";; 3) st="--steps 20 --warmup 3";; 4) st="--steps 10 --warmup 2";; 5) st="--steps 20 --warmup 3";; esac
  ( time timeout 300 python bench.py --config $c $st --no-pmc --no-cpu-baseline ) > gpurun_out/r5w/bench_config$c.json 2> gpurun_out/r5w/bench$c.err
  cp gpurun_out/step_trace_config$c/steady_step.txt gpurun_out/r5w/steady_step_config$c.txt 2>/dev/null
done
rm -rf gpurun_out/step_trace_config*/ gpurun_out/pmc_config*/ gpurun_out/step_pmc_config*/
for c in 1 2 3 4 5; do python - $c <<'PY'
import json, sys
c = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r5w/bench_config{c}.json").read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(c, d["value"], d["unit"], d["ms_per_step"], "frac", r.get("frac"), "clock", d.get("sustained_shader_clock_ghz_each"))
except Exception as e:
    print(c, "FAILED", e)
PY
done
grep real gpurun_out/r5w/*.err
