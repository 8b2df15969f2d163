#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box variation on the pool is ~2 %, more than most kernel
# tweaks are worth): alternates bench runs between csrc/libdrgnn_old.so and csrc/libdrgnn.so through DRGNN_LIB.
#   cp deeprank-gnn_amd/csrc/libdrgnn.so deeprank-gnn_amd/csrc/libdrgnn_old.so   # the baseline build
#   ... edit, make -C deeprank-gnn_amd/csrc ...
#   gpurun -- 'bash tools/ab_bench.sh [rounds] [nets...]'
# Prints per run: library, net, us per step, [step+topo, update, topo alone, step alone] kernel us (GINet), final loss
# (identical losses = the change did not touch the arithmetic).
R=${1:-3}
shift
NETS=${@:-GINet}
for i in $(seq 1 $R); do
for lib in libdrgnn_old.so libdrgnn.so; do
for net in $NETS; do
DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/$lib python bench.py --no-cpu-baseline --epoch-graphs 0 --net $net 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$lib $net',round(d['ms_per_step']*1000,2),[round(v['avg_us'],2) for v in k.values()],d['config']['final_loss'])"
done; done; done
