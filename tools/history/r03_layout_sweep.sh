#!/bin/bash
# GINet: one vs two workgroups per graph (forced) and the automatic choice, over batch sizes, topology rebuilt / cached
OUT=${1:-gpurun_out/layout}; mkdir -p $OUT
for mode in cached rebuilt; do for b in 64 128 256 1024; do for lay in two seq one auto; do
  python bench.py --net GINet --topology $mode --graphs-per-gpu $b --step-layout $lay --no-cpu-baseline --epoch-graphs 0 --min-seconds 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read())
print('GINet $mode B=$b layout=$lay  %.2f us/step  %.3f M graphs/s  loss %s' % (d['ms_per_step']*1000, d['value']/1e6, d['config']['final_loss']))" | tee -a $OUT/layout_sweep.txt
done; done; done
