#!/bin/bash
# where does the one-rank RCCL bench spend its time?  faulthandler dumps every thread's stack after T seconds and exits
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp1
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for g in 1 0; do
echo "== DRGNN_DP_GRAPH=$g"
DRGNN_DP_GRAPH=$g timeout 120 python -c "
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(${DP1_T:-60}, exit=True)
sys.argv = ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5', '--force-dp-path', '--no-cpu-baseline', '--epoch-graphs', '0', '--no-other-nets']
runpy.run_path('bench.py', run_name='__main__')
" > gpurun_out/dp1/trace_out_$g.txt 2> gpurun_out/dp1/trace_err_$g.txt < /dev/null
echo rc=$?; tail -c 400 gpurun_out/dp1/trace_out_$g.txt; grep -v "Warning\|warn\|amdgpu.ids" gpurun_out/dp1/trace_err_$g.txt | head -60
done
