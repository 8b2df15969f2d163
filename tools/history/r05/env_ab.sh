#!/bin/bash
# same-box A/B of an environment setting under the driver's bench arguments: tools/r05/env_ab.sh "<VAR=value>" [net] [rounds]
cd $GRAFT_REPO_ROOT
V=$1; NET=${2:-GINet}; R=${3:-2}
O=gpurun_out/r05_env; mkdir -p $O
for r in $(seq 1 $R); do
  for mode in base env; do
    if [ $mode = env ]; then E="$V"; else E="DRGNN_NOOP=1"; fi
    env $E timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --net $NET --no-cpu-baseline --no-other-nets --min-seconds 2 > $O/$mode.$NET.$r.json 2>/dev/null < /dev/null
    echo "== $mode ($E) $NET round $r"; python tools/r05/bench_brief.py $O/$mode.$NET.$r.json | grep -v "k_topo\|without"
  done
done
