#!/bin/bash
# in-step sensitivity to the deferred-rescale threshold of the attention kernel (FLUXMI_ATTN_THR), alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3j
for r in 1 2; do
  for t in 8 40; do
    FLUXMI_ATTN_THR=$t timeout 300 python bench.py --steps 28 --warmup 3 --no-pmc --no-cpu-baseline > gpurun_out/r3j/b_${t}_$r.json 2> gpurun_out/r3j/b_${t}_$r.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/r3j/b_${t}_$r.json').read().strip().splitlines()[-1]); print('THR=${t} run $r:', d['value'], 'it/s', d['ms_per_step'], 'ms')
PY
  done
done
