cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_fullsize.py tests/test_gpu_epoch.py tests/test_gpu_trainer.py -m gpu -x -q > $O/pytest.log 2>&1
tail -n 12 $O/pytest.log
for net in GINet sGAT FoutNet; do
python bench.py --no-cpu-baseline --epoch-graphs 0 --min-seconds 1 --net $net 2>$O/err_$net.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$net', 'us/step', round(d['ms_per_step']*1000,2), 'kernels', [round(v['avg_us'],2) for v in k.values()], 'loss', d['config']['final_loss'])" | tee -a $O/ab.txt
done
