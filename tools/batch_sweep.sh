#!/bin/bash
# graphs/s of the three nets at batch 64 / 256 / 1024, topology rebuilt every step and cached (DESIGN.md's table)
mkdir -p gpurun_out/sweep
for net in GINet sGAT FoutNet; do for mode in rebuilt cached; do for b in ${SWEEP_BATCHES:-64 128 256 1024}; do
  python bench.py --net $net --topology $mode --graphs-per-gpu $b --no-cpu-baseline --epoch-graphs 0 --min-seconds ${SWEEP_SECONDS:-3} 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read())
print('$net $mode B=$b  %.2f us/step  %.3f M graphs/s' % (d['ms_per_step']*1000, d['value']/1e6))" | tee -a gpurun_out/sweep/sweep.txt
done; done; done
