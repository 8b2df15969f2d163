#!/bin/bash
# Round-5 batch sweep (three nets x rebuilt / cached x batch 64..1024) and the reference's shipped-model shape
# (48 features, batch 128: SURVEY 6) through bench.py; results under gpurun_out/r05s/
mkdir -p gpurun_out/r05s gpurun_out/sweep
rm -f gpurun_out/sweep/sweep.txt
SWEEP_SECONDS=${SWEEP_SECONDS:-2} timeout 420 bash tools/batch_sweep.sh < /dev/null > /dev/null 2>&1
cp gpurun_out/sweep/sweep.txt gpurun_out/r05s/batch_sweep.txt
for net in GINet sGAT FoutNet; do for mode in rebuilt cached; do
  timeout 60 python bench.py --net $net --topology $mode --n-feat 48 --graphs-per-gpu 128 --no-cpu-baseline --epoch-graphs 0 --no-other-nets \
     --min-seconds 2 < /dev/null 2>/dev/null > gpurun_out/r05s/f48_b128_${net}_${mode}.json
done; done
ls -la gpurun_out/r05s
cat gpurun_out/r05s/batch_sweep.txt
