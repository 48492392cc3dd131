#!/bin/bash
# round-2 GPU batch B: whole GPU suite, full-geometry parity, attention variants + PMC, bench for every config (+ live PMC for config 2)
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== attention A/B"
timeout 300 python tools/attn_ab.py --rounds 5 --iters 20 --L 4608 2816 2>&1 | tee $O/attn_ab.txt
echo "== GPU suite (no full geometry)"
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -k "not full_geometry and not full_depth" -p no:cacheprovider > $O/pytest_ops.log 2>&1; echo "rc=$?"; tail -12 $O/pytest_ops.log | cut -c1-220
echo "== full-geometry parity (2+2 cases)"
timeout 900 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "teacher_forced" -p no:cacheprovider > $O/pytest_full22.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|  -- |passed|failed|Error" $O/pytest_full22.log | cut -c1-250 | tail -120
echo "== bench config 2 with live PMC"
timeout 900 python bench.py --steps 28 --warmup 3 --pmc > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?"; cut -c1-700 $O/bench_c2.json; tail -3 $O/bench_c2.err
for c in 3 5 1; do
  echo "== bench config $c"
  timeout 600 python bench.py --steps 28 --warmup 3 --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err; echo "rc=$?"; cut -c1-600 $O/bench_c$c.json; tail -2 $O/bench_c$c.err
done
echo "== attention PMC"
FLUXMI_ATTN_VAR=4 timeout 400 bash tools/pmc_attn.sh $O/pmc_attn > $O/pmc_attn.log 2>&1; tail -40 $O/pmc_attn.log
echo "== rocprofv3 kernel trace of the steady state"
( cd /tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err ); echo "rc=$?"
python tools/rocprof_summary.py $O/prof --steady -o $O/rocprof_steady_step.txt --header "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline   (MI355X, round 2, batch B)" > $O/rocprof_summary.log 2>&1; head -12 $O/rocprof_steady_step.txt
echo "== full depth 19+38"
timeout 1200 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "full_depth" -p no:cacheprovider > $O/pytest_full57.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|  -- |passed|failed|Error" $O/pytest_full57.log | cut -c1-250 | tail -40
