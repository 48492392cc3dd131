mkdir -p gpurun_out/r5r
cd $GRAFT_REPO_ROOT
for v in base pf20 pf40 base pf20 pf40; do
  if [ $v = base ]; then unset FLUXMI_LIB; else export FLUXMI_LIB=$GRAFT_REPO_ROOT/flux-fp8-api_amd/fluxmi/variants/libfluxmi_$v.so; fi
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-step-trace --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['ms_per_step_each'], d['sustained_shader_clock_ghz_each'])" >> gpurun_out/r5r/pf_cap.log
done
cat gpurun_out/r5r/pf_cap.log
