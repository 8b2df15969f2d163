#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats of the recorded drop-in loop (tools/r06/dropin_trace.py), one run per
# variant; only the stats tables are kept (the traces are tens of MB).   usage: dropin_profile.sh OUTDIR NET BATCH [variants...]
OUT=$1; NET=$2; B=$3; shift 3
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $OUT; OUT=$(cd $OUT && pwd)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  set -- ${v//_/ }
  d=/tmp/prof_${NET}_${B}_$v
  rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --stats -d $d -o run --output-format csv -- python $ROOT/tools/r06/dropin_trace.py $NET $B $1 $2 > $OUT/${NET}_b${B}_$v.log 2>&1 < /dev/null
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${NET}_b${B}_${v}_kernel_stats.csv
  grep "us per step" $OUT/${NET}_b${B}_$v.log
done
