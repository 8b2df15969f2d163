#!/bin/bash
# Same-box comparison of several builds of the library: bash tools/ab_libs.sh <rounds> <net> lib1.so lib2.so ...
# (files under deeprank-gnn_amd/csrc/, selected through DRGNN_LIB; must share the C ABI of the checked-out Python side)
R=$1; NET=$2; shift 2
for i in $(seq 1 $R); do
for lib in "$@"; do
DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/$lib python bench.py --no-cpu-baseline --epoch-graphs 0 --net $NET 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$lib $NET',round(d['ms_per_step']*1000,2),[round(v['avg_us'],2) for v in k.values()])"
done; done
