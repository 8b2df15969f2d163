#!/bin/bash
# In-loop kernel durations (rocprofv3 --kernel-trace --stats of the default bench) for one or more library builds:
#   bash tools/kstats.sh <net> lib1.so [lib2.so ...]      (files under deeprank-gnn_amd/csrc/)
export TMPDIR=/tmp
NET=$1; shift
for lib in "$@"; do
  rm -rf /tmp/kst; mkdir -p /tmp/kst
  DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/$lib rocprofv3 --kernel-trace --stats -d /tmp/kst -o run --output-format csv -- python bench.py --net $NET --no-cpu-baseline --epoch-graphs 0 > /tmp/kst/out.log 2>&1
  echo "== $lib $NET: $(grep -o '"ms_per_step": [0-9.]*' /tmp/kst/out.log | head -1)"
  f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
  head -4 $f | cut -d, -f1-4,6-7
done
