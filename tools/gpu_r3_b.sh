#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/r3b; mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
L=gpurun_out/r3b/tl.log
for a in 16 23 31; do for b in 2; do
  echo "abl $a bias $b" >> $L
  FLUXMI_SK_BIAS=$b FLUXMI_SK_ABL=$a timeout 200 python tools/sk_timeline.py >> $L 2>&1
done; done
grep -v amdgpu.ids $L
