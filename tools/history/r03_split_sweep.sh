#!/bin/bash
# builder workgroups per graph chosen by residency of the whole launch (libdrgnn.so) vs always two up to 160 graphs (libdrgnn_x.so),
# batch sizes around the resident limit, topology rebuilt every step
for net in GINet sGAT FoutNet; do for b in 64 72 80 96 112 128 144 160 192; do for lib in libdrgnn_x.so libdrgnn.so; do
  DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/$lib python bench.py --net $net --graphs-per-gpu $b --no-cpu-baseline --epoch-graphs 0 --min-seconds 0.5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read())
print('$net rebuilt B=$b $lib  %.2f us/step  %.3f M graphs/s' % (d['ms_per_step']*1000, d['value']/1e6))"
done; done; done
