#!/bin/bash
# round 3, last GPU call: attention tests on the final sources, then the default bench run (live PMC keyed by the final source hash)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3last
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -2
( time timeout 900 python bench.py ) > gpurun_out/r3last/bench_c2.json 2> gpurun_out/r3last/bench_c2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3last/bench_c2.json').read().strip().splitlines()[-1]); print(d["value"], d["unit"], d["ms_per_step"], 'roofline', d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("mfma_busy_frac_pmc"), d["roofline"]["source"])
PY
