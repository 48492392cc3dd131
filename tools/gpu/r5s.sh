mkdir -p gpurun_out/r5s
cd $GRAFT_REPO_ROOT
timeout 300 python tools/attn_timeline.py --L 4608 --split 0,2 > gpurun_out/r5s/timeline_4608.log 2>&1
timeout 300 python tools/attn_timeline.py --L 2816 --split 0,1 > gpurun_out/r5s/timeline_2816.log 2>&1
grep -v amdgpu gpurun_out/r5s/timeline_4608.log gpurun_out/r5s/timeline_2816.log | cut -c1-400
