cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout 600 python tools/r04/check_step2.py GINet 2>&1 | grep -v "^$" > $O/check.log
cat $O/check.log
for r in 1 2; do
for lay in old auto; do
python bench.py --no-cpu-baseline --epoch-graphs 0 --min-seconds 1 --net GINet --step-layout $lay 2>$O/err_$lay.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$lay GINet', 'us/step', round(d['ms_per_step']*1000,2), 'kernels', [round(v['avg_us'],2) for v in k.values()], 'loss', d['config']['final_loss'])" | tee -a $O/ab.txt
done; done
for lib in libdrgnn_prof.so libdrgnn_prof8.so; do
  [ -f deeprank-gnn_amd/csrc/$lib ] && PROF_LIB=$lib timeout 300 python tools/r04/topo_phases.py 0 > $O/topo_$lib.txt 2>&1
  [ -f deeprank-gnn_amd/csrc/$lib ] && PROF_LIB=$lib timeout 300 python tools/r04/topo_phases.py 1 > $O/topo_w_$lib.txt 2>&1
done
tail -n 60 $O/topo_*.txt
