#!/bin/bash
# round-2 GPU batch J: GEMM epilogues -- batched bias / gate / residual loads, one kernel per hot epilogue (no accumulator spills)
O=gpurun_out/${OUT:-r02j}; mkdir -p $O
export TMPDIR=/tmp
echo "== GEMM / engine correctness"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest_ops.log 2>&1; echo "rc=$?"; tail -6 $O/pytest_ops.log | cut -c1-200
echo "== bench A/B (separate processes): specialised epilogue kernels | run-time switch kernels | specialised"
for v in "esel1:" "esel0:FLUXMI_GEMM_ESEL=0" "esel1b:"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 400 python bench.py --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$? $(python -c "
import json; d=json.loads(open('$O/bench_$n.json').read().strip().split(chr(10))[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['achieved'], [(l['launch'].split('(')[0], l['us']) for l in r['launches']])" 2>&1 | tail -1)"
done
echo "== full-geometry parity (tiny + c2)"
timeout 900 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "teacher_forced and (tiny or c2)" -p no:cacheprovider > $O/pytest_full22.log 2>&1; echo "rc=$?"; grep -E "  BAD|passed|failed|Error" $O/pytest_full22.log | cut -c1-260 | tail -8
