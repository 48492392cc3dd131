#!/bin/bash
# round-2 GPU batch C: attention variants (deeper V prefetch, MFMA row sums), full-geometry parity
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
echo "== attention A/B"
timeout 300 python tools/attn_ab.py --rounds 5 --iters 20 --L 4608 2816 2>&1 | tee $O/attn_ab.txt
echo "== attention + ln tests"
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "attention or ln_modulate" -p no:cacheprovider > $O/pytest_attn.log 2>&1; echo "rc=$?"; grep -E "L=|passed|failed|Error|assert" $O/pytest_attn.log | tail -15
echo "== full-geometry parity (tiny harness + 2+2 cases)"
timeout 900 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "teacher_forced" -p no:cacheprovider > $O/pytest_full22.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|  -- |passed|failed|Error" $O/pytest_full22.log | cut -c1-260 | tail -160
echo "== full depth 19+38"
timeout 1200 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -k "full_depth" -p no:cacheprovider > $O/pytest_full57.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|  -- |passed|failed|Error" $O/pytest_full57.log | cut -c1-260 | tail -40
