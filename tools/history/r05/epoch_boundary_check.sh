#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_epoch.py tests/test_gpu_trainer.py tests/test_gpu_dp8.py -x -q -m gpu < /dev/null 2>&1 | tail -4
bash tools/r05/epoch_host_breakdown.sh
bash tools/r05/epoch_boundary.sh 2>&1 | grep -v "^rep\|host time"
