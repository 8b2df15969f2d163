#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak; mkdir -p $O
for i in 1 2; do
  s=$(date +%s)
  timeout 280 python -m pytest tests -x -q -m gpu -p no:cacheprovider -o faulthandler_timeout=100 < /dev/null 2>&1 | tail -60 > $O/pipe_$i.log
  echo "piped run $i $(( $(date +%s) - s )) s: $(tail -1 $O/pipe_$i.log | cut -c1-150)"
  ps -eo pid,etimes,cmd | grep -i "python" | grep -v grep | cut -c1-150
done
