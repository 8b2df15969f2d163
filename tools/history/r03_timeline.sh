#!/bin/bash
# Cumulative timeline of the fused step kernel: libraries built with -DDRGNN_EXIT_AFTER=k (tools/ablate_step.sh build "base exit1 ...")
# timed for each net.  usage: tools/r03_timeline.sh <outdir> [nets...]
OUT=${1:-gpurun_out/timeline}; shift
NETS=${@:-GINet sGAT FoutNet}
mkdir -p $OUT
for net in $NETS; do
  for v in base exit1 exit2 exit3 exit4 exit5 exit6 exit8 exit9 exit10 exit11 exit12 exit14 exit15; do
    f=deeprank-gnn_amd/csrc/ablate/libdrgnn_$v.so
    [ -f $f ] || continue
    DRGNN_LIB=$f timeout 120 python tools/time_step.py $v $net 2>/dev/null | grep "^skip" >> $OUT/timeline_$net.txt
  done
done
cat $OUT/timeline_*.txt
