for i in 1 2; do
for lib in libdrgnn_old.so libdrgnn.so; do
for b in 64 256; do
DRGNN_LIB=$PWD/deeprank-gnn_amd/csrc/$lib python bench.py --no-cpu-baseline --epoch-graphs 0 --net GINet --graphs-per-gpu $b 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$lib B=$b',round(d['ms_per_step']*1000,2),round(d['value']/1e6,3),[round(v['avg_us'],2) for v in k.values()],d['config']['final_loss'])"
done; done; done
