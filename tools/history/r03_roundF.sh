#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
bash tools/ab.sh 3 "GINet sGAT FoutNet" "" libdrgnn_a.so libdrgnn.so 2>&1 | tee $OUT/ab_tn.txt
