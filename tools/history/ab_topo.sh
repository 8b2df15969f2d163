for r in 1 2; do for n in sGAT GINet FoutNet; do for lib in variants/libdrgnn_pre_topo.so libdrgnn.so; do
DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/$lib python bench.py --net $n --no-cpu-baseline --epoch-graphs 0 --min-seconds 2 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k = r['roofline']['kernels']
print('$n', '$lib', 'us/step %.3f' % (r['ms_per_step'] * 1e3), '| k_topo %.2f' % [v['avg_us'] for kk, v in k.items() if kk.startswith('k_topo')][0], '| step+topo %.2f' % k[r['roofline']['kernel']]['avg_us'])
"
done; done; done
