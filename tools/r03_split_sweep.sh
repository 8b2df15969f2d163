#!/bin/bash
# experiment: builder with two workgroups per graph (default up to 160 graphs) vs one, batch sizes around the resident limit
for net in GINet sGAT FoutNet; do for b in 64 96 128 160 192 256; do for sp in 160 0; do
  DRGNN_TOPO_SPLIT_MAX=$sp DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/libdrgnn_x.so python bench.py --net $net --graphs-per-gpu $b --no-cpu-baseline --epoch-graphs 0 --min-seconds 0.5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read())
print('$net rebuilt B=$b split<=$sp  %.2f us/step  %.3f M graphs/s' % (d['ms_per_step']*1000, d['value']/1e6))"
done; done; done
