#!/bin/bash
# One gpurun call: GPU test-suite, the bench line under the driver's arguments, profile collection for the nets given.
# usage: tools/gpu_round.sh <tag> [nets...]
TAG=${1:-r}
shift
NETS=${@:-GINet sGAT FoutNet}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -5 gpurun_out/$TAG/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver.json 2> gpurun_out/$TAG/bench_driver.err
tail -c 1500 gpurun_out/$TAG/bench_driver.json
for net in $NETS; do
  timeout 1200 bash tools/collect_profiles.sh $TAG/$net $net > gpurun_out/$TAG/collect_$net.log 2>&1
  tail -c 600 gpurun_out/$TAG/$net/benchline_driver_args.json
done
