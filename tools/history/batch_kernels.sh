#!/bin/bash
# per-kernel times of the step at larger batches (measurement tool): tools/batch_kernels.sh "<nets>" "<batches>" [topology]
NETS=${1:-GINet}; BS=${2:-"256 1024"}; TOPO=${3:-cached}
for n in $NETS; do for b in $BS; do
python bench.py --net $n --graphs-per-gpu $b --topology $TOPO --no-cpu-baseline --epoch-graphs 0 --min-seconds 2 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k = r['roofline']['kernels']
print('$n B=$b $TOPO: us/step %.2f  graphs/s %.0f |' % (r['ms_per_step'] * 1e3, r['value']), ' | '.join('%s %.2f' % (kk.split(' ')[0][:24], v['avg_us']) for kk, v in k.items()))
"
done; done
