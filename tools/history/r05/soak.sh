#!/bin/bash
# the GPU suite several times in a row on one box, with a stack dump of a test that takes longer than 100 s
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak; mkdir -p $O
for i in 1 2 3 4; do
  s=$(date +%s)
  timeout 270 python -m pytest tests -x -q -m gpu -o faulthandler_timeout=100 < /dev/null > $O/run_$i.log 2>&1
  echo "run $i rc=$? $(( $(date +%s) - s )) s: $(tail -1 $O/run_$i.log | cut -c1-150)"
done
