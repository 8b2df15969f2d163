#!/bin/bash
OUT=gpurun_out/r03e; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
bash tools/ab.sh 3 "GINet" "" libdrgnn_m0.so libdrgnn_m1.so 2>&1 | tee $OUT/ab_merge.txt
timeout 200 python tools/phase_timing_step.py GINet > $OUT/phase_GINet.log 2>&1; tail -50 $OUT/phase_GINet.log
