# round 5, call k: thin last round folded into the round in front of it (attention, 768^2) -- parity, isolated, in-step
mkdir -p gpurun_out/r5k
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -s -k "attention" ) > gpurun_out/r5k/pytest_attn.log 2>&1
echo "rc=$?" >> gpurun_out/r5k/pytest_attn.log
( time timeout 600 python -m pytest tests/test_full_geometry_gpu.py tests/test_engine_gpu.py -m gpu -q -x -s -k "c3_2p2 or tiny or fused_equals" ) > gpurun_out/r5k/pytest_engine.log 2>&1
echo "rc=$?" >> gpurun_out/r5k/pytest_engine.log
timeout 300 python tools/attn_probe.py --split 0,1 --L 2816 > gpurun_out/r5k/attn_probe.log 2>&1
timeout 300 python tools/attn_probe.py --split 0,1 --L 1536 >> gpurun_out/r5k/attn_probe.log 2>&1
timeout 300 python tools/attn_probe.py --split 0,1 --L 2048 >> gpurun_out/r5k/attn_probe.log 2>&1
timeout 300 python tools/attn_probe.py --split 0,1 --L 3072 >> gpurun_out/r5k/attn_probe.log 2>&1
timeout 600 python tools/ab_step.py --height 768 --width 768 --embedders --rounds 3 --variant default: --variant nosplit:attn_split=0 > gpurun_out/r5k/ab_768.log 2>&1
timeout 600 python tools/ab_step.py --height 512 --width 512 --rounds 3 --variant default: --variant nosplit:attn_split=0 > gpurun_out/r5k/ab_512.log 2>&1
( time python bench.py --config 3 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline ) > gpurun_out/r5k/bench_config3.json 2> gpurun_out/r5k/bench_config3.err
cp gpurun_out/step_trace_config3/steady_step.txt gpurun_out/r5k/steady_step_config3.txt 2>/dev/null
rm -rf gpurun_out/step_trace_config*/
tail -n 4 gpurun_out/r5k/pytest_attn.log gpurun_out/r5k/pytest_engine.log; grep -v amdgpu gpurun_out/r5k/attn_probe.log; grep -v amdgpu gpurun_out/r5k/ab_768.log gpurun_out/r5k/ab_512.log; head -c 250 gpurun_out/r5k/bench_config3.json
