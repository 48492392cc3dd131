mkdir -p gpurun_out/r5o
cd $GRAFT_REPO_ROOT
timeout 900 python tools/ab_step.py --rounds 4 --check --variant prefetch1:prefetch=1 --variant prefetch2:prefetch=2 --variant prefetch3:prefetch=3 > gpurun_out/r5o/ab_prefetch.log 2>&1
timeout 600 python tools/ab_step.py --height 768 --width 768 --embedders --rounds 3 --variant prefetch1:prefetch=1 --variant prefetch2:prefetch=2 --variant prefetch3:prefetch=3 > gpurun_out/r5o/ab_prefetch_768.log 2>&1
grep -v amdgpu gpurun_out/r5o/ab_prefetch.log gpurun_out/r5o/ab_prefetch_768.log
