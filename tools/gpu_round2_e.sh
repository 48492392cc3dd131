#!/bin/bash
# round-2 GPU batch E: fp16-K folded attention (+ lagged consumers), streaming LayerNorm as default; parity at real geometry with them
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
echo "== attention A/B"
timeout 400 python tools/attn_ab.py --rounds 5 --iters 20 --L 4608 2816 8192 2>&1 | tee $O/attn_ab.txt
echo "== attention + ln + qkv_rope tests"
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "attention or ln_modulate or qkv_rope" -p no:cacheprovider > $O/pytest_attn.log 2>&1; echo "rc=$?"; grep -E "L=|raw-Q|passed|failed|Error|assert" $O/pytest_attn.log | tail -15
echo "== bench A/B (separate processes): defaults | lagged consumers | bf16 K (unfolded)"
for v in "base:" "lag:FLUXMI_ATTN_VAR=1" "bf16k:FLUXMI_ATTN_F16K=0" "base2:"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 400 python bench.py --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$? $(python -c "import json,sys; d=json.loads(open('$O/bench_$n.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_hipevent'), d['roofline']['achieved'], d['roofline']['attention']['achieved'])" 2>&1 | tail -1)"
done
echo "== full-geometry parity (tiny harness + c2), shipping kernels"
timeout 900 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "teacher_forced and (tiny or c2)" -p no:cacheprovider > $O/pytest_full22.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|  -- |passed|failed|Error" $O/pytest_full22.log | cut -c1-260 | grep -i "attention\|whole block\|end to end\|gate\|passed\|failed\|BAD" | tail -60
echo "== engine tests"
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_engine.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_engine.log
