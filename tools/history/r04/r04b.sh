cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 600 python tools/r04/check_step2.py FoutNet sGAT > gpurun_out/r04b/check.log 2>&1
tail -30 gpurun_out/r04b/check.log
