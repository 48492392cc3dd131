# round 5, call t: FINAL sources -- whole GPU suite + smoke
mkdir -p gpurun_out/r5t
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x -s --durations=12 ) > gpurun_out/r5t/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5t/pytest_full.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r5t/smoke.log 2>&1
grep -E "passed|failed|rc=" gpurun_out/r5t/pytest_full.log | tail -n 3; grep smoke gpurun_out/r5t/smoke.log
