#!/bin/bash
# rocprofv3 --kernel-trace --stats of the step beyond the resident batch size (GINet one workgroup per graph, branches side by side)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/bigk; mkdir -p $O
for cfg in "32 128 rebuilt" "32 256 cached" "32 1024 cached" "48 128 rebuilt"; do set -- $cfg
  rm -rf /tmp/bk; mkdir -p /tmp/bk
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/bk -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --net GINet --n-feat $1 --graphs-per-gpu $2 --topology $3 --no-cpu-baseline --epoch-graphs 0 --no-other-nets --min-seconds 1 > /tmp/bk/out.log 2>&1 < /dev/null)
  echo "== GINet F=$1 B=$2 $3: $(grep -o '"ms_per_step": [0-9.]*' /tmp/bk/out.log | head -1)"
  f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -4 "$f" | cut -d, -f1-4,6-7
done
