# round 5, call h: host-op probe on the GPU box's CPU (oracle drift), third bench lease, config 3 / 4 second lease
mkdir -p gpurun_out/r5h
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OMP_NUM_THREADS=32 python tools/probes/host_ops_probe.py > gpurun_out/r5h/probe_box_default.txt 2>&1
OMP_NUM_THREADS=32 ONEDNN_MAX_CPU_ISA=AVX512_CORE_BF16 python tools/probes/host_ops_probe.py > gpurun_out/r5h/probe_box_capped.txt 2>&1
OMP_NUM_THREADS=8 ONEDNN_MAX_CPU_ISA=AVX512_CORE python tools/probes/host_ops_probe.py > gpurun_out/r5h/probe_box_avx512core_t8.txt 2>&1
lscpu | head -30 > gpurun_out/r5h/lscpu.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc ) > gpurun_out/r5h/bench_driver_cmd_nopmc.json 2> gpurun_out/r5h/bench.err
cp gpurun_out/step_trace_config2/steady_step.txt gpurun_out/r5h/steady_step_config2.txt 2>/dev/null
( time python bench.py --config 3 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline ) > gpurun_out/r5h/bench_config3.json 2> gpurun_out/r5h/bench_config3.err
( time python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/r5h/bench_config4.json 2> gpurun_out/r5h/bench_config4.err
rm -rf gpurun_out/step_trace_config*/ gpurun_out/pmc_config*/
paste -d'|' gpurun_out/r5h/probe_box_default.txt gpurun_out/r5h/probe_box_capped.txt | cut -c1-150
head -c 200 gpurun_out/r5h/bench_driver_cmd_nopmc.json; echo; head -c 200 gpurun_out/r5h/bench_config3.json; echo; head -c 300 gpurun_out/r5h/bench_config4.json
