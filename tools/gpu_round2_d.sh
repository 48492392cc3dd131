#!/bin/bash
# round-2 GPU batch D: attention fold / prologue / store variants, streaming LayerNorm, concurrent hybrid peel, parity with the new gates
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
echo "== attention A/B"
timeout 400 python tools/attn_ab.py --rounds 5 --iters 20 --L 4608 2816 8192 2>&1 | tee $O/attn_ab.txt
echo "== LN A/B"
timeout 200 python tools/ln_ab.py 2>&1 | tee $O/ln_ab.txt
echo "== attention + ln tests"
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "attention or ln_modulate" -p no:cacheprovider > $O/pytest_attn.log 2>&1; echo "rc=$?"; grep -E "L=|passed|failed|Error|assert" $O/pytest_attn.log | tail -15
echo "== bench A/B (separate processes): base | concurrent peel | attention fold+prologue | base"
for v in "base:" "hybrid2:FLUXMI_GEMM_HYBRID=2" "attn5:FLUXMI_ATTN_VAR=5" "ln1:FLUXMI_LN_V=1" "base2:"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 400 python bench.py --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$? $(python -c "import json,sys; d=json.loads(open('$O/bench_$n.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_hipevent'), d['roofline']['achieved'])" 2>&1 | tail -1)"
done
echo "== full-geometry parity (tiny harness + 2+2 cases), shipping kernels"
timeout 900 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "teacher_forced" -p no:cacheprovider > $O/pytest_full22.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|  -- |passed|failed|Error" $O/pytest_full22.log | cut -c1-260 | tail -120
echo "== same, attention fold (FLUXMI_ATTN_VAR=5): tiny + c2"
FLUXMI_ATTN_VAR=5 timeout 600 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "teacher_forced and (tiny or c2)" -p no:cacheprovider > $O/pytest_full22_fold.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|  -- |passed|failed|Error" $O/pytest_full22_fold.log | cut -c1-260 | grep -i "attn\|block\|passed\|failed\|BAD" | tail -60
echo "== full depth 19+38"
timeout 1200 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "full_depth" -p no:cacheprovider > $O/pytest_full57.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|  -- |passed|failed|Error" $O/pytest_full57.log | cut -c1-260 | tail -30
