cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r04g/pytest.log 2>&1
tail -25 gpurun_out/r04g/pytest.log
