cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
timeout 1800 python -m pytest tests/test_gpu_epoch.py tests/test_gpu_trainer.py -m gpu -q > gpurun_out/r04h/pytest.log 2>&1
tail -n 25 gpurun_out/r04h/pytest.log
