#!/bin/bash
# full GPU suite + the sweep rows a layout / build change touches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05v
timeout 1500 python -m pytest tests -x -q -m gpu < /dev/null 2>&1 | tail -5 | tee gpurun_out/r05v/pytest.txt
SWEEP_BATCHES="${SWEEP_BATCHES:-128 256}" MODE=rebuilt timeout 300 bash tools/r05/ab_batch128.sh "" "sGAT FoutNet" "128 256"
MODE=cached timeout 300 bash tools/r05/ab_batch128.sh "" "sGAT FoutNet" "128"
