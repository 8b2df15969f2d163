#!/bin/bash
# the round's final evidence in ONE gpurun call: GPU suite, per-net profiles (NAME, default r05_v3), batch sweep, 48-feature batch-128 lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
timeout 900 python -m pytest tests -x -q -m gpu < /dev/null 2>&1 | tail -3 > gpurun_out/r05f/pytest.txt
for net in GINet sGAT FoutNet; do timeout 1200 bash tools/r05/final_collect.sh $net ${NAME:-r05_v3} > gpurun_out/r05f/final_$net.log 2>&1; done
timeout 700 bash tools/r05/sweep_and_f48.sh > gpurun_out/r05f/sweep.log 2>&1
cat gpurun_out/r05f/pytest.txt; ls gpurun_out/r05f/out | wc -l; tail -3 gpurun_out/r05f/sweep.log
