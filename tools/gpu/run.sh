#!/bin/bash
# One parametrised entry for every gpurun call (replaces the per-call r5?.sh scripts of round 5):
#   gpurun --timeout S -- 'bash tools/gpu/run.sh <tag> <recipe> [recipe ...]'
# Every recipe writes under gpurun_out/<tag>/<n>_* and prints a short tail; summaries worth judging are copied to profiles/ afterwards.
#   suite            the driver's command: pytest -x -q -m gpu (+ durations), then smoke()
#   suite_hog        the same with a CPU hog on half of the host's hardware threads beside it (a "busy box")
#   bench            the driver's bench command (python bench.py --gpus 1, defaults) with its in-step trace
#   bench_pmc        bench.py --gpus 1 --pmc-step (separate counter passes)
#   bench:<args>     bench.py with explicit arguments:  bench:--config_3_--steps_20     ('_' -> space, '%' -> '_')
#   py:<script+args> python <script> <args>:            py:tools/attn%probe.py_--L_4608  ('_' -> space, '%' -> '_')
#   pytest:<k-expr>  pytest -m gpu -x -q -s -k <expr>:  pytest:attention+or+qkv_rope     ('+' -> space)
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
words() { echo "$1" | tr '_' ' ' | tr '%' '_'; }
keep_traces() {
  for d in gpurun_out/step_trace_config*; do
    [ -d "$d" ] && cp "$d/steady_step.txt" "$out/${1}_steady_step_$(basename "$d" | sed s/step_trace_//).txt" 2>/dev/null
  done
  for f in gpurun_out/step_pmc_config*/*.json; do [ -f "$f" ] && cp "$f" "$out/${1}_$(basename "$f")"; done
  rm -rf gpurun_out/step_trace_config*/ gpurun_out/pmc_config*/ gpurun_out/step_pmc_config*/
}
n=0
for r in "$@"; do
  n=$((n+1))
  case "$r" in
    suite|suite_hog)
      log=$out/${n}_pytest_gpu.log
      : > "$log"
      hog=""
      if [ "$r" = suite_hog ]; then
        python tools/gpu/cpu_hog.py --fraction 0.5 --seconds 1500 & hog=$!
        echo "cpu hog (pid $hog) spinning on half of $(nproc) hardware threads beside the suite" >> "$log"
      fi
      ( time python -m pytest tests -x -q -m gpu --durations=15 ) >> "$log" 2>&1; echo "pytest rc=$?" >> "$log"
      [ -n "$hog" ] && kill "$hog" 2>/dev/null
      ( time python -c "import __graft_entry__ as g; g.smoke()" ) > "$out/${n}_smoke.log" 2>&1; echo "smoke rc=$?" >> "$out/${n}_smoke.log"
      grep -E "passed|failed|rc=|real" "$log" | tail -n 4; grep -E "smoke|rc=" "$out/${n}_smoke.log" | tail -n 3 ;;
    bench|bench_pmc|bench:*)
      args="--gpus 1"
      [ "$r" = bench_pmc ] && args="--gpus 1 --pmc-step"
      case "$r" in bench:*) args=$(words "${r#bench:}") ;; esac
      ( time python bench.py $args ) > "$out/${n}_bench.json" 2> "$out/${n}_bench.err"; echo "bench rc=$?" >> "$out/${n}_bench.err"
      keep_traces "$n"
      head -c 400 "$out/${n}_bench.json"; echo; tail -n 2 "$out/${n}_bench.err" ;;
    py:*)
      ( time timeout 1200 python $(words "${r#py:}") ) > "$out/${n}_py.log" 2>&1; echo "rc=$?" >> "$out/${n}_py.log"
      grep -v amdgpu "$out/${n}_py.log" | tail -n 40 ;;
    pytest:*)
      k=$(echo "${r#pytest:}" | tr '+' ' ')
      ( time timeout 1500 python -m pytest tests -m gpu -x -q -s -k "$k" ) > "$out/${n}_pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/${n}_pytest.log"
      grep -E "passed|failed|rc=|Error|assert" "$out/${n}_pytest.log" | tail -n 12 ;;
    *) echo "unknown recipe $r"; exit 2 ;;
  esac
done
