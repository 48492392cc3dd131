#!/bin/bash
# round-2 GPU batch A: op tests + attention A/B + GEMM A/B + bench + kernel trace, then the full-geometry parity tests.
# Usage on the GPU box (via gpurun): bash tools/gpu_round2_a.sh
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
echo "== rocm-smi"; rocm-smi --showproductname 2>/dev/null | head -8; nproc; free -g | head -2
echo "== op + engine tests (not full geometry)"
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -x -k "not full_geometry and not full_depth" -p no:cacheprovider > $O/pytest_ops.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_ops.log
echo "== new tests verbose"
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -q -s -k "deferred or production or batch_sharded" -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "rc=$?"; grep -E "L=|group|without|passed|failed|Error|error" $O/pytest_new.log | tail -40
echo "== attention A/B"
timeout 300 python tools/attn_ab.py --rounds 5 --iters 20 2>&1 | tee $O/attn_ab.txt
echo "== gemm A/B"
for sh in lin1 qkv mlp0; do timeout 200 python tools/gemm_probe.py --shape $sh --ab 13,8,16,4 --rounds 5 2>&1 | tee -a $O/gemm_ab.txt; done
timeout 200 python tools/gemm_probe.py --shape lin2 --ab 16,13,8 --rounds 5 2>&1 | tee -a $O/gemm_ab.txt
echo "== bench config 2"
timeout 600 python bench.py --steps 28 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?"; cut -c1-1500 $O/bench_c2.json; tail -3 $O/bench_c2.err
echo "== bench round-1 attention kernel for comparison"
FLUXMI_ATTN_V=1 timeout 300 python bench.py --steps 28 --warmup 3 --no-cpu-baseline > $O/bench_c2_attnv1.json 2> $O/bench_c2_attnv1.err; cut -c1-400 $O/bench_c2_attnv1.json
echo "== rocprofv3 kernel trace of the steady state"
R=${GRAFT_REPO_ROOT:-$(pwd)}
( cd /tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err ); echo "rc=$?"
python tools/rocprof_summary.py $O/prof --steady -o $O/rocprof_steady_step.txt --header "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline   (MI355X, round 2)" > $O/rocprof_summary.log 2>&1; head -30 $O/rocprof_steady_step.txt; tail -3 $O/rocprof_summary.log
echo "== full-geometry parity"
timeout 1500 python -m pytest tests/test_full_geometry_gpu.py -m gpu -q -s -p no:cacheprovider > $O/pytest_full.log 2>&1; echo "rc=$?"; grep -E "^\[|  ok |  BAD|passed|failed|Error" $O/pytest_full.log | cut -c1-230 | tail -150
