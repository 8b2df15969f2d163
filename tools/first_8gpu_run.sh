#!/bin/bash
# First run on a multi-GPU node: one table over the data-parallel schedules bench.py knows, each run
# preceded by its own --dp-selftest (all-reduced gradient == rank 0's recompute on the union of the shards, parameters in sync).
#   rows   : 1 / 2 / 4 / 8 GPUs
#   columns: RCCL all-reduce recorded inside the hipGraph (default) | DRGNN_DP_ONESHOT=1 (one-shot peer-to-peer all-reduce,
#            adopted only after its verified trial) | DRGNN_DP_GRAPH=0 (eager all-reduce between graph replays)
# Every run also times the DISTINCT-MINI-BATCH replay per rank (`--dp-distinct`: a cycle of 32 different synthetic mini-batches
# per rank through the same data-parallel schedule), so that the first real curve is not a curve of L2-resident replays
# (VERDICT r04 item 8), and the table FAILS (exit 1) when a multi-rank RCCL run reports rccl_ranks != gpus or parameters out of sync.
# usage: bash tools/first_8gpu_run.sh [net] [max gpus] [backend]   (run from the repository root; writes gpurun_out/first_8gpu/)
#        backend gloo: the plumbing check on ONE GPU (ranks time-share it): bash tools/first_8gpu_run.sh GINet 2 gloo
NET=${1:-GINet}
MAXG=${2:-8}
BACKEND=${3:-nccl}
OUT=gpurun_out/first_8gpu; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # run <tag> <gpus> <env...>
  local tag=$1 n=$2; shift 2
  local port=$((29500 + RANDOM % 2000))
  if [ "$n" = 1 ]; then
    env "$@" timeout 600 python bench.py --gpus 1 --net $NET --backend $BACKEND --dp-selftest --dp-distinct --no-cpu-baseline --epoch-graphs 0 > $OUT/${tag}_$n.json 2> $OUT/${tag}_$n.err
  else
    env "$@" timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $n --net $NET --backend $BACKEND --dp-selftest --dp-distinct --no-cpu-baseline --epoch-graphs 0 > $OUT/${tag}_$n.json 2> $OUT/${tag}_$n.err
  fi
  echo "rc=$?" >> $OUT/${tag}_$n.err
}
for n in 1 2 4 8; do
  [ $n -le $MAXG ] || continue
  run rccl_graph $n DRGNN_DP_GRAPH=1
  run oneshot $n DRGNN_DP_ONESHOT=1
  run eager $n DRGNN_DP_GRAPH=0
done
FIRST_RUN_BACKEND=$BACKEND python - <<'PY'
import glob, json, os
out = "gpurun_out/first_8gpu"
import sys
backend = os.environ.get("FIRST_RUN_BACKEND", "nccl")
bad = []
print("%-12s %5s %14s %10s %12s %-28s %10s %14s" % ("schedule", "gpus", "graphs/s", "us/step", "distinct us", "dp_exchange", "rccl_ranks", "params_in_sync"))
for tag in ("rccl_graph", "oneshot", "eager"):
    for n in (1, 2, 4, 8):
        f = os.path.join(out, "%s_%d.json" % (tag, n))
        if not os.path.exists(f):
            continue
        line = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if not line:
            print("%-12s %5d  FAILED (see %s)" % (tag, n, f.replace(".json", ".err")))
            bad.append("%s_%d: no bench line" % (tag, n))
            continue
        d = json.loads(line[-1]); c = d.get("config", {})
        st = c.get("dp_selftest") or {}
        ranks = st.get("rccl_ranks", c.get("rccl_ranks"))
        dd = (d.get("dp_distinct") or {}).get("us_per_step")
        print("%-12s %5d %14.0f %10.2f %12s %-28s %10s %14s" % (tag, n, d["value"], d["ms_per_step"] * 1e3,
                                                            "%.2f" % dd if dd else "-", str(c.get("dp_exchange"))[:28], ranks,
                                                            c.get("params_in_sync")))
        if n > 1 and c.get("params_in_sync") is not True:
            bad.append("%s_%d: parameters not in sync over the ranks" % (tag, n))
        if n > 1 and backend == "nccl" and ranks != n:
            bad.append("%s_%d: rccl_ranks = %s, expected %d (the collective did not span the ranks)" % (tag, n, ranks, n))
if bad:
    print("FIRST RUN FAILED:\n  " + "\n  ".join(bad))
    sys.exit(1)
PY
