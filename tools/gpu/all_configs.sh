# all five BASELINE.json configs back to back on one lease (gpurun -- bash tools/gpu/all_configs.sh): one summary line per config, JSON lines under gpurun_out/r6x/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6x
for c in 1 2 3 4 5; do
  python bench.py --config $c --steps 20 --warmup 3 --no-pmc --no-cpu-baseline > gpurun_out/r6x/config$c.json 2> gpurun_out/r6x/config$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r6x/config$c.json'))
r=d['roofline']
print('config $c: %.2f %s  %.3f ms/step  clock %s  roofline %s frac %.4f  step %s' % (d['value'], d['unit'], d['ms_per_step'], d['sustained_shader_clock_ghz_each'], r.get('bound'), r.get('frac') or 0, r.get('step')))
PY
done
rm -rf gpurun_out/step_trace_config*
