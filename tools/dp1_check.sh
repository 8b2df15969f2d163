#!/bin/bash
# the data-parallel schedule of bench.py with ONE rank over RCCL (what a 1-GPU box can check of the driver's multi-GPU runs):
# plain launch and under torch.distributed.run.  (A fresh box pages torch in for a minute or two: warmed up first.)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp1
python -c "import torch; print('torch in', torch.__version__)"
A="--gpus 1 --steps 20 --warmup 5 --force-dp-path --dp-selftest --no-cpu-baseline --epoch-graphs 0 --no-other-nets"
echo "== python bench.py $A"
DRGNN_BENCH_WATCHDOG=150 timeout 200 python bench.py $A > gpurun_out/dp1/plain.txt 2> gpurun_out/dp1/plain.err < /dev/null
echo "rc=$?"; tail -c 900 gpurun_out/dp1/plain.txt; grep -v "Warning\|warn\|amdgpu.ids\|socket.cpp" gpurun_out/dp1/plain.err | tail -12
echo "== torch.distributed.run --nproc-per-node 1 bench.py $A"
DRGNN_BENCH_WATCHDOG=150 timeout 220 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py $A \
   > gpurun_out/dp1/run.txt 2> gpurun_out/dp1/run.err < /dev/null
echo "rc=$?"; tail -c 900 gpurun_out/dp1/run.txt; grep -v "Warning\|warn\|amdgpu.ids\|socket.cpp" gpurun_out/dp1/run.err | tail -12
