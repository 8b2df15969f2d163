#!/bin/bash
# like r03_timeline.sh, but timed inside hipGraph replays (tools/time_graph.py: no host launch floor); GINet only
OUT=${1:-gpurun_out/timeline}; mkdir -p $OUT
for v in base exit1 exit2 exit3 exit4 exit5 exit6 exit8 exit9 exit10 exit11 exit12 exit14 exit15; do
  f=deeprank-gnn_amd/csrc/ablate/libdrgnn_$v.so
  [ -f $f ] || continue
  DRGNN_LIB=$f timeout 120 python tools/time_graph.py $v 2>/dev/null | grep "^graph" >> $OUT/timeline_graph_GINet.txt
done
cat $OUT/timeline_graph_GINet.txt
