#!/bin/bash
# same-box A/B of library variants at larger batches, topology rebuilt: tools/r05/ab_batch128.sh "<variant.so ...>" "<nets>" "<batches>"
cd $GRAFT_REPO_ROOT
for r in 1 2; do for net in ${2:-sGAT}; do for b in ${3:-128}; do for lib in base $1; do
  if [ $lib = base ]; then L=$PWD/deeprank-gnn_amd/csrc/libdrgnn.so; else L=$PWD/deeprank-gnn_amd/csrc/variants/$lib; fi
  DRGNN_LIB=$L timeout 100 python bench.py --net $net --topology ${MODE:-rebuilt} --graphs-per-gpu $b --no-cpu-baseline --epoch-graphs 0 --no-other-nets --min-seconds 2 < /dev/null 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read())
print('$lib $net B=$b round $r  %.2f us/step  kernel %.2f' % (d['ms_per_step']*1000, d['roofline']['kernel_us']))"
done; done; done; done
