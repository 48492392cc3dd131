#!/bin/bash
# round-2 GPU batch L (final, after the GEMM epilogue rework): whole GPU suite, bench per config (config 2 with live PMC), rocprofv3 steady-step trace, attention PMC, smoke
O=gpurun_out/r02l; mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?"; tail -3 $O/smoke.log
echo "== bench config 2 with live PMC"
timeout 900 python bench.py --steps 28 --warmup 3 --pmc > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?"; cut -c1-600 $O/bench_c2.json; tail -3 $O/bench_c2.err
for c in 3 5 1; do
  echo "== bench config $c"
  timeout 600 python bench.py --steps 28 --warmup 3 --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err; echo "rc=$?"; cut -c1-400 $O/bench_c$c.json; tail -2 $O/bench_c$c.err
done
echo "== rocprofv3 kernel trace of the steady state"
( cd /tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err ); echo "rc=$?"
python tools/rocprof_summary.py $O/prof --steady -o $O/rocprof_steady_step.txt --header "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline   (MI355X, round 2, final batch)" > $O/rocprof_summary.log 2>&1; head -14 $O/rocprof_steady_step.txt
echo "== GPU suite"
timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_all.log | cut -c1-220
