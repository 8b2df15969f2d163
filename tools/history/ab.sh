#!/bin/bash
# Same-box A/B of library builds (files under deeprank-gnn_amd/csrc/, selected through DRGNN_LIB; they must share the C ABI
# of the checked-out Python side).  usage: tools/ab.sh <rounds> "<nets>" "<extra bench args>" lib1.so lib2.so ...
R=$1; NETS=$2; EXTRA=$3; shift 3
for i in $(seq 1 $R); do
for net in $NETS; do
for lib in "$@"; do
DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/$lib python bench.py --no-cpu-baseline --epoch-graphs 0 --min-seconds 1 --net $net $EXTRA 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$lib $net', 'us/step', round(d['ms_per_step']*1000,2), 'kernels', [round(v['avg_us'],2) for v in k.values()], 'loss', d['config']['final_loss'])"
done; done; done
