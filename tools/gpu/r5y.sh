# round 5, call y: bf16 tile configs, bits
mkdir -p gpurun_out/r5y
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 300 python tools/probes/bf16_cfg_bits_probe.py ) > gpurun_out/r5y/probe.log 2>&1
echo "rc=$?" >> gpurun_out/r5y/probe.log
tail -n 60 gpurun_out/r5y/probe.log
