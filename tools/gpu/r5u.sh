# round 5, call u: serving soak on the final sources (mixed resolutions / batch sizes through one engine)
mkdir -p gpurun_out/r5u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 400 python tools/soak.py --cycles 4 ) > gpurun_out/r5u/soak.log 2>&1
echo "soak rc=$?" >> gpurun_out/r5u/soak.log
tail -n 12 gpurun_out/r5u/soak.log
