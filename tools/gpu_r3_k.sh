#!/bin/bash
# in-step A/B on one box: default 8-wave attention kernel vs its mid-barrier variant (FLUXMI_ATTN_V=3), alternating
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3k
for r in 1 2; do
  for v in 0 3; do
    if [ $v = 0 ]; then unset FLUXMI_ATTN_V; else export FLUXMI_ATTN_V=$v; fi
    timeout 300 python bench.py --steps 28 --warmup 3 --no-pmc --no-cpu-baseline > gpurun_out/r3k/b_${v}_$r.json 2> gpurun_out/r3k/b_${v}_$r.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/r3k/b_${v}_$r.json').read().strip().splitlines()[-1]); print('ATTN_V=${v} run $r:', d['value'], 'it/s', d['ms_per_step'], 'ms')
PY
  done
done
