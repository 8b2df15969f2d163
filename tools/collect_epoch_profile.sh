#!/bin/bash
# Runs on the GPU box (gpurun): whole-epoch throughput of the three loop variants for the three nets + a rocprofv3
# kernel-trace of the GINet run.   usage: tools/collect_epoch_profile.sh <tag>  -> gpurun_out/<tag>/
set -e
TAG=${1:-epoch}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for net in GINet sGAT FoutNet; do
  python tools/epoch_bench.py --graphs 4096 --epochs 3 --net $net > $OUT/epoch_$net.jsonl 2> $OUT/epoch_$net.err
done
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python tools/epoch_bench.py --graphs 4096 --epochs 3 --only native-epoch > $OUT/stats.log 2>&1
head -12 $OUT/stats/run_kernel_stats.csv
cat $OUT/epoch_*.jsonl | cut -c1-200
