#!/bin/bash
# A/B inside the step on one box: attention4 (default) vs attention2 folded (FLUXMI_ATTN_V=2), alternating
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3i
for r in 1 2 3; do
  for v in 0 2; do
    if [ $v = 0 ]; then unset FLUXMI_ATTN_V; else export FLUXMI_ATTN_V=2; fi
    timeout 300 python bench.py --steps 28 --warmup 3 --no-pmc --no-cpu-baseline > gpurun_out/r3i/b_${v}_$r.json 2> gpurun_out/r3i/b_${v}_$r.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/r3i/b_${v}_$r.json').read().strip().splitlines()[-1]); print('ATTN_V=${v} run $r:', d['value'], 'it/s', d['ms_per_step'], 'ms; attention', d.get('attention',{}).get('us'), 'us', d.get('attention',{}).get('kernel','')[:20])
PY
  done
done
