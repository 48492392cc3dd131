#!/bin/bash
# round 3, GPU batch A: stream-K GEMM (config 17) correctness + A/B against config 16 / 13, and its effect on the step
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/r3a; mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "stream_k or (test_f8_gemm and 17-)" > gpurun_out/r3a/pytest_sk.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a/pytest_sk.log
tail -5 gpurun_out/r3a/pytest_sk.log
for shp in lin2 4608,3072,12288 4608,3072,3072; do
  timeout 200 python tools/gemm_probe.py --shape $shp --epi gate --ab 16,17 --rounds 5 >> gpurun_out/r3a/ab.log 2>&1
done
for b in 0 1 3 4 6; do echo "bias $b" >> gpurun_out/r3a/ab.log; FLUXMI_SK_BIAS=$b timeout 200 python tools/gemm_probe.py --shape lin2 --epi gate --ab 16,17 --rounds 3 >> gpurun_out/r3a/ab.log 2>&1; done
cat gpurun_out/r3a/ab.log
FLUXMI_GEMM_SK=0 timeout 400 python bench.py --steps 14 --warmup 3 --no-cpu-baseline > gpurun_out/r3a/bench_sk0.json 2> gpurun_out/r3a/bench_sk0.err
timeout 400 python bench.py --steps 14 --warmup 3 --no-cpu-baseline > gpurun_out/r3a/bench_sk1.json 2> gpurun_out/r3a/bench_sk1.err
python - <<'PY'
import json
for n in ("sk0","sk1"):
    try:
        d=json.loads(open(f"gpurun_out/r3a/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], [(r["launch"], r["us"]) for r in d["roofline"]["launches"]])
    except Exception as e:
        print(n, "failed", e)
PY
