cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
PROF_LIB=libdrgnn_prof.so timeout 300 python tools/r04/topo_phases.py 0 7 2>&1 | grep -v amdgpu | tee $O/topo_lean.txt
PROF_LIB=libdrgnn_prof.so timeout 300 python tools/r04/topo_phases.py 1 7 2>&1 | grep -v amdgpu | tee $O/topo_lean_w.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_fullsize.py tests/test_gpu_epoch.py -m gpu -x -q > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout 600 python tools/r04/time_topo.py 64 128 256 2>&1 | grep k_topo | tee $O/time_topo.txt
for net in GINet sGAT FoutNet; do
python bench.py --no-cpu-baseline --epoch-graphs 0 --min-seconds 1 --net $net 2>$O/err_$net.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$net', 'us/step', round(d['ms_per_step']*1000,2), 'kernels', [round(v['avg_us'],2) for v in k.values()], 'loss', d['config']['final_loss'])" | tee -a $O/ab.txt
done
