#!/bin/bash
# cumulative timeline of the fused step kernel (libraries built with -DDRGNN_EXIT_AFTER=k: tools/ablate_step.sh build "base exit1 ..."),
# timed inside hipGraph replays (tools/time_graph.py: no host launch floor).  usage: tools/r03_timeline_graph.sh <outdir> [nets...]
OUT=${1:-gpurun_out/timeline}; shift; mkdir -p $OUT
NETS=${@:-GINet}
for net in $NETS; do
for v in base exit1 exit2 exit3 exit4 exit5 exit6 exit8 exit9 exit10 exit11 exit12 exit14 exit15; do
  f=deeprank-gnn_amd/csrc/ablate/libdrgnn_$v.so
  [ -f $f ] || continue
  DRGNN_LIB=$f timeout 120 python tools/time_graph.py $v $net 2>/dev/null | grep "^graph" >> $OUT/timeline_graph_$net.txt
done
cat $OUT/timeline_graph_$net.txt
done
