#!/bin/bash
# same-box timing of library variants with the merged update schedule: tools/ab_variants.sh <rounds> "<nets>" lib1 lib2 ...
R=$1; NETS=$2; shift 2
for r in $(seq 1 $R); do
  for n in $NETS; do
    for lib in "$@"; do
      DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/$lib python bench.py --net $n --no-cpu-baseline --epoch-graphs 0 --min-seconds 2 ${EXTRA_ARGS} 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$n', '$lib', 'us/step %.3f' % (r['ms_per_step'] * 1e3), '| loss', r['config']['final_loss'])
"
    done
  done
done
