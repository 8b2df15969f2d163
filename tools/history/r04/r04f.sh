cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
timeout 600 python tools/r04/check_step2.py FoutNet sGAT 2>&1 | grep -v "eager" > gpurun_out/r04f/check.log
cat gpurun_out/r04f/check.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04f/pytest.log 2>&1
tail -15 gpurun_out/r04f/pytest.log
