#!/bin/bash
# First run on a multi-GPU node: one table over the data-parallel schedules bench.py knows, each run
# preceded by its own --dp-selftest (all-reduced gradient == rank 0's recompute on the union of the shards, parameters in sync).
#   rows   : 1 / 2 / 4 / 8 GPUs
#   columns: RCCL all-reduce recorded inside the hipGraph (default) | DRGNN_DP_ONESHOT=1 (one-shot peer-to-peer all-reduce,
#            adopted only after its verified trial) | DRGNN_DP_GRAPH=0 (eager all-reduce between graph replays)
# usage: bash tools/first_8gpu_run.sh [net] [max gpus]      (run from the repository root; writes gpurun_out/first_8gpu/)
NET=${1:-GINet}
MAXG=${2:-8}
OUT=gpurun_out/first_8gpu; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # run <tag> <gpus> <env...>
  local tag=$1 n=$2; shift 2
  local port=$((29500 + RANDOM % 2000))
  if [ "$n" = 1 ]; then
    env "$@" python bench.py --gpus 1 --net $NET --dp-selftest --no-cpu-baseline --epoch-graphs 0 > $OUT/${tag}_$n.json 2> $OUT/${tag}_$n.err
  else
    env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $n --net $NET --dp-selftest --no-cpu-baseline --epoch-graphs 0 > $OUT/${tag}_$n.json 2> $OUT/${tag}_$n.err
  fi
  echo "rc=$?" >> $OUT/${tag}_$n.err
}
for n in 1 2 4 8; do
  [ $n -le $MAXG ] || continue
  run rccl_graph $n DRGNN_DP_GRAPH=1
  run oneshot $n DRGNN_DP_ONESHOT=1
  run eager $n DRGNN_DP_GRAPH=0
done
python - <<'PY'
import glob, json, os
out = "gpurun_out/first_8gpu"
print("%-12s %5s %14s %10s %-28s %10s %14s" % ("schedule", "gpus", "graphs/s", "us/step", "dp_exchange", "rccl_ranks", "params_in_sync"))
for tag in ("rccl_graph", "oneshot", "eager"):
    for n in (1, 2, 4, 8):
        f = os.path.join(out, "%s_%d.json" % (tag, n))
        if not os.path.exists(f):
            continue
        line = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if not line:
            print("%-12s %5d  FAILED (see %s)" % (tag, n, f.replace(".json", ".err")))
            continue
        d = json.loads(line[-1]); c = d.get("config", {})
        print("%-12s %5d %14.0f %10.2f %-28s %10s %14s" % (tag, n, d["value"], d["ms_per_step"] * 1e3, c.get("dp_exchange"),
                                                       c.get("rccl_ranks"), c.get("params_in_sync")))
PY
